// gv_kernels.hpp — host-visible launch interface of the gfx950 kernels (internal to libgarden_vis).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gv {

// ---- device mirror of the component pools (HBM layout) ----
// Streams are split so that a kernel reads only what the pool needs: 65 B per entity for a flat pool whose
// meshes and transforms pair up 1:1 (TRS 40 + flags 1 + AABB 24: the algorithmic minimum), +4 B parent entry
// with a hierarchy, +4 B transform entry when mesh and transform pools are ordered independently.
constexpr uint32_t kSlotMask = 0x0FFFFFFFu;
constexpr uint32_t kSlotNone = 0x0FFFFFFFu;
constexpr uint32_t kXfActive = 1u;         // selfActive && ancestorsActive (transform.hpp:110)
constexpr uint32_t kXfWithAncestors = 2u;  // modelWithAncestors (transform.hpp:60)
constexpr uint32_t kXfLive = 4u;           // entity != 0
constexpr uint32_t kMeshCandidate = 1u << 28;  // entity != 0 && isEnabled (mesh.cpp:142), in MeshMirror::link

// One 32-byte record per transform entry: {pos.xyz, scale.x | quat xyzw}. Kept together because every reader wants
// both halves: the streaming kernels read the pair at full rate (round-1 probe tools/stream_probe.hip, in the history: 6.8 TB/s interleaved vs
// 6.7 TB/s as two arrays) and every gather (emit, ancestor walks) pays one 64-byte sector instead of two.
struct XfAB {
    float4 a;  // (pos.x, pos.y, pos.z, scale.x)
    float4 b;  // quat xyzw
};
struct TransformMirror {
    const XfAB* ab;
    const float2* c;         // (scale.y, scale.z)
    const uint8_t* flags;    // kXf* bits
    const unsigned long long* active_bits;  // bit e of word e/64 = kXfActive of entry e (derived from flags[] on the
                                            // device after every upload); lets a flat, exactly-paired pool fetch its
                                            // 64 flags with one wave-uniform 8-byte load instead of a byte per lane
    const uint32_t* parent;  // parent entry or kSlotNone; read only when max_depth > 0
    uint32_t count;
    uint32_t max_depth;  // longest parent chain (bounds the walk; a cycle is rejected at mirror build)
};
enum MeshMapping : uint32_t {
    kMapGeneral = 0,    // read link[], then the transform entry it names
    kMapSpeculate = 1,  // >= 90 % of the entries map to their own index: prefetch xf[i] beside mesh[i], verify with link[]
    kMapExact = 2       // every candidate maps to its own index: link[] is not read at all
};
struct MeshMirror {
    const float4* a;       // (aabb.min.xyz, aabb.max.x)
    const float2* b;       // (aabb.max.y, aabb.max.z); non-candidate entries hold an empty box (min == max == 0)
    const uint32_t* link;  // transform entry | kMeshCandidate
    uint32_t count;
    uint32_t mapping;      // MeshMapping, decided at mirror build (speed only: every mapping is handled correctly)
    const uint32_t* orig;  // mirror entry -> pool slot (null: the mirror is in pool order)
};

struct HizDevice {
    const float* depth;          // mip 0
    const float2* mips;          // levels >= 1, (min,max); rg16f: the same texels as packed binary16 pairs (uint32_t each)
    const uint64_t* mip_offset;  // device array [GV_MAX_MIPS], in texels
    uint32_t width, height, mip_count;
    uint32_t level1_virtual;  // level 1 is not stored (even sizes, fused build): a query derives its texels from the depth
    uint32_t rg16f;   // GV_CONFIG_HIZ_RG16F: texels are RG16F, min rounded toward -inf / max toward +inf (level 0 stays the depth)
    uint32_t nested;  // every level's min bounds ALL texels it covers (even sizes all the way, or the conservative rule):
                      // a coarser level may then prove occlusion early (exact shortcut, see hiz_occluded)
};

constexpr uint32_t kCullBlock = 256;   // slots per cull workgroup (4 waves)
constexpr uint32_t kFusedEmitMaxSlots = 32768;  // pools up to this size cull + emit in one launch: the look-back chain costs
                                                // ~16 ns per tile, so it only pays while a launch costs more (6.9 vs 9.8 us at 10 k
                                                // slots, equal at 100 k, 64 vs 24 us at 1 M; profiles/r02i_fused_emit.txt)
constexpr uint32_t kAutoBoundsMinSlots = 262144;  // pools above this size get block bounds unless GV_CONFIG_LINEAR_SCAN (smaller
                                                  // ones are launch-bound and keep the one-launch / batched paths)
constexpr uint32_t kEmitChunk = 4096;  // slots per compaction chunk = 64 ballot words
constexpr uint32_t kEmitParts = 4;     // emit workgroups per chunk: 1024 slots = 16 ballot words each

struct ViewParams {
    float planes[6][4];
    uint32_t plane_count;
    float cam[3];
    float cam_offset[3];
    float vp[16];
    uint32_t write_is_visible;  // main pass. The cull kernels store the bytes only when no emit follows the launch (count-only
                                // views, the one-launch cull + emit): with records requested the emit kernel expands them from
                                // the ballot words it reads anyway, as whole sectors — the byte stores cost the bandwidth-bound
                                // cull kernel 5-7 us of 105 at 10 M entities for 1.5 % of its bytes (round-3 probe tools/read_probe.hip, in the history)
    uint32_t use_hiz;
    uint32_t distance_2d;
};

struct ViewBuffers {
    unsigned long long* mask;  // one ballot word per 64 slots
    uint32_t* chunk_count;     // visible per kEmitChunk slots (atomically summed by the cull workgroups, re-zeroed by scan)
    uint32_t* chunk_count_next;  // the other of the two alternating totals buffers (cleared by the self-prefixing emit)
    uint32_t* chunk_offset;    // exclusive scan of chunk_count (scan path only)
    uint32_t* draw_count;      // total
    uint8_t* is_visible;       // per slot (main pass)
    uint8_t* vis_flags;        // per emit workgroup (kEmitParts per chunk): 1 = its quarter of the chunk's isVisible bytes may hold
                               // a non-zero, 0 = known to be all zero. Lets the self-prefixing emit leave an EMPTY chunk after one
                               // load (behind an occlusion pass most chunks are empty). NULL: every workgroup writes its bytes
    uint32_t* visible_idx;     // compact records [0, draw_count), ascending slot order
    float* baked_model;        // 12 floats per record
    float* distance_sq;
};

// Emit seeds (round 3): for a flat, exactly paired pool that is at rest, everything the emit kernel gathers per visible entry —
// TRS record, scale tail, pool slot: three 64-byte sectors for 44 useful bytes, and the sparse gather is what bounds the emit
// behind an occlusion pass (316 k records -> 60 MB of sectors at cfg3) — as ONE 64-byte record per mirror entry, built on the
// device while the pool's mirror is clean (like the block bounds; a pool that changes every frame goes without). Same bits.
struct EmitSeed {
    float4 a, b;      // XfAB
    float2 c;         // (scale.y, scale.z)
    uint32_t orig;    // pool slot of the entry
    uint32_t pad[5];
};
static_assert(sizeof(EmitSeed) == 64, "one sector per visible entry");
constexpr uint32_t kEmitSeedMinSlots = 262144;  // smaller pools are launch-bound (and take other emit paths)
hipError_t launch_emit_seeds(const MeshMirror& mesh, const TransformMirror& xf, EmitSeed* seeds, hipStream_t stream);

// Block bounds (opt-in, GV_CONFIG_BLOCK_BOUNDS): world-space AABB of all corners of the candidates of each 256-entry
// cull workgroup, built while the mirror is clean. A workgroup whose box lies behind one frustum plane by more than
// the rounding margin skips its streams: every entity in it would have failed that plane in the per-entity test.
struct BlockBounds {
    const float4* lo = nullptr;   // xyz = min corner (+inf when the block has no candidate; -inf when a member is non-finite);
                                  // w = the largest sphere reach of its candidates (block_window; +inf likewise)
    const float4* hi = nullptr;   // xyz = max corner (-inf / +inf likewise)
    uint8_t* examined = nullptr;  // per workgroup: 1 = ran the per-entity path, 0 = skipped (statistics)
};
hipError_t launch_block_bounds(const MeshMirror& mesh, const TransformMirror& xf, float4* lo, float4* hi, hipStream_t stream);
// ... kept current while a few entries change per frame: flags (one byte per 256-entry block, padded to a multiple of 16, set by
// launch_mark_dirty_blocks for every block that holds a re-mirrored entry) name the blocks whose box — and whose entries' emit
// seeds, when `seeds` is given — are re-derived; the flags are cleared.
hipError_t launch_block_patch(const MeshMirror& mesh, const TransformMirror& xf, float4* lo, float4* hi, EmitSeed* seeds, uint8_t* flags,
                              hipStream_t stream);
// the dirty slot ranges of a sync (start[nranges + 1]: running slot counts, first[nranges]: first slots; device memory) -> flags
hipError_t launch_mark_dirty_blocks(const uint32_t* start, const uint32_t* first, uint32_t nranges, uint32_t total, const uint32_t* inv,
                                    uint32_t slots, uint32_t entries, uint8_t* flags, hipStream_t stream);
hipError_t launch_cull(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz, const ViewParams& vp,
                       const ViewBuffers& out, hipStream_t stream);
// Block bounds as classify (+ window test for Hi-Z views) + cull over the listed workgroups (two or three launches; same outputs
// as a cull of every workgroup). kept_count / next_count: two alternating device counters (both zero before the first use; each
// launch clears the other one for the next); kept_list: cull_list_entry_bytes() per workgroup; kept_flag: a byte per workgroup.
size_t cull_list_entry_bytes();
hipError_t launch_cull_listed(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz, const ViewParams& vp,
                              const ViewBuffers& out, const BlockBounds& bounds, uint32_t* kept_count, uint32_t* next_count,
                              void* kept_list, uint8_t* kept_flag, hipStream_t stream);
// Cull + order-stable compaction + record emission of one view in ONE launch (decoupled look-back over the 256-entry
// tiles; gv_cull.hip): same outputs as launch_cull + launch_emit except mask / chunk counts, which it does not produce.
// status: one 64-bit word per tile (zero-initialised once, never cleared: words carry `epoch`); ticket: a running
// counter whose value before this launch is ticket_base.
hipError_t launch_cull_emit(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz, const ViewParams& vp,
                            const ViewBuffers& out, unsigned long long* status, uint32_t* ticket, uint32_t ticket_base, uint32_t epoch,
                            hipStream_t stream);
// One pass over the streams for up to kMaxBatchViews views that share views[0].cam (Hi-Z only on view 0).
constexpr uint32_t kMaxBatchViews = 8;
struct MultiViewPlanes {
    float planes[6][4];
    uint32_t plane_count;
    uint32_t write_is_visible;
};
hipError_t launch_cull_multi(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz,
                             const ViewParams* views, const ViewBuffers* outs, uint32_t nviews, hipStream_t stream,
                             const BlockBounds* bounds = nullptr);
// read-only pass over the cull kernel's input streams (65 B per entry); gv_debug_stream_peak
hipError_t launch_stream_probe(const MeshMirror& mesh, const TransformMirror& xf, float* sink, hipStream_t stream);
// Table-driven tick (gv_cull_batch_begin): the culls of several small pools in one launch, their emits in one launch.
// The descriptor structs are private to gv_cull.hip: the host fills a table of entries through these.
size_t cull_table_entry_bytes();
void fill_cull_table_entry(void* entry, const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz,
                           const ViewParams* views, const ViewBuffers* outs, uint32_t nviews);
hipError_t launch_cull_table(const void* device_table, uint32_t jobs, uint32_t max_slots, hipStream_t stream);
size_t emit_table_entry_bytes();
void fill_emit_table_entry(void* entry, const MeshMirror& mesh, const TransformMirror& xf, const ViewParams& vp,
                           const ViewBuffers& out, uint32_t clear_chunks, const float4* world);
hipError_t launch_emit_table(const void* device_table, uint32_t entries, uint32_t max_slots, hipStream_t stream);
hipError_t launch_scan(const ViewBuffers& out, uint32_t chunk_count, hipStream_t stream);
// self_prefix: every emit workgroup derives its chunk's base from the chunk totals itself (no launch_scan in front;
// pools of up to kSelfPrefixMaxChunks chunks); out.chunk_count / chunk_count_next then alternate from cull to cull.
constexpr uint32_t kSelfPrefixMaxChunks = 4096;
// world: the resident world matrices of the current transform mirror (3 float4 per entry), or NULL — records then take
// world[slot] instead of re-walking the parent chain (same bits)
hipError_t launch_emit(const MeshMirror& mesh, const TransformMirror& xf, const ViewParams& vp, const ViewBuffers& out,
                       hipStream_t stream, bool self_prefix = false, uint32_t clear_chunks = 0, const float4* world = nullptr,
                       const EmitSeed* seeds = nullptr);
// all views of a batched cull in one launch (self-prefixing form; views[v] / outs[v] / clear_chunks[v] per view)
hipError_t launch_emit_batch(const MeshMirror& mesh, const TransformMirror& xf, const ViewParams* views, const ViewBuffers* outs,
                             const uint32_t* clear_chunks, uint32_t nviews, hipStream_t stream, const float4* world = nullptr);
// map: pool slot -> caller's global id (NULL: identity), applied before `base`
hipError_t launch_copy_idx(const uint32_t* src, const uint32_t* count, uint32_t* dst, uint32_t capacity, uint32_t base,
                           const uint32_t* map, hipStream_t stream);
hipError_t launch_copy_shard(const uint32_t* src, const uint32_t* count, uint32_t* dst, uint32_t capacity, uint32_t base,
                             const uint32_t* map, hipStream_t stream);
// gv_exchange_views: every list of a frame into ONE shard [n + total, c_0 .. c_{n-1}, list 0, list 1 ...] (gv_shard.hip)
struct ShardItem {
    const uint32_t* src;    // the view's visible_idx
    const uint32_t* count;  // its device draw count
    const uint32_t* map;    // pool slot -> caller's global id (NULL: identity)
    uint32_t base, capacity;
};
hipError_t launch_copy_shard_batch(const ShardItem* device_items, uint32_t n, uint32_t widest, uint32_t* dst, hipStream_t stream);
// gv_exchange_visible / _views: the leading hdr_words words of all `world` gathered rows (the true counts; the per-list counts of
// a batched frame), then `seq`, into pinned host memory (host_words[1 + r * hdr_words + w] = rows[r * row_words + w], host_words[0]
// = seq with system-scope release): the host learns the counts of a frame without an event record or a synchronisation — it
// looks at the word a frame or two later
hipError_t launch_exchange_headers(const uint32_t* rows, uint32_t row_words, uint32_t world, uint32_t hdr_words, uint32_t* host_words, uint32_t seq,
                                   hipStream_t stream);
// GV_EXCHANGE_PEER: the 1 + shard[0] leading words of a rank's staging shard (at most cap_words) into its row of EVERY rank's rows
// (rows.dst[r]: this rank's row in rank r's buffer — the rank's own, or a peer device's over xGMI), one launch
struct PeerRows {
    uint32_t* dst[64];  // GV_EXCHANGE_MAX_RANKS (garden_vis.h; asserted in gv_exchange.cpp)
};
hipError_t launch_peer_scatter(const uint32_t* shard, uint32_t cap_words, const PeerRows& rows, uint32_t world, hipStream_t stream);
// gv_results_fetch of a pool of up to kPublishMaxSlots slots: device results -> pinned host buffers in one launch
// (up to kPublishLdsSlots slots the isVisible bytes are put back into pool-slot order in LDS by the same kernel)
constexpr uint32_t kPublishMaxSlots = 262144;
constexpr uint32_t kPublishLdsSlots = 32768;
// the caller's record struct (GvRecordLayout); stride == 0: none
struct RecordLayout {
    uint32_t stride, component_offset, baked_model, distance_sq, buffer_index, component_stride, buffer_index_value;
    uint32_t pad_;
    const uint32_t* slot_map;  // GV_RESULTS_MAP_RECORDS: componentOffset = slot_map[slot] * component_stride (NULL: slot itself); filled in
                               // when results are delivered, never kept in PoolState::record_layout
};
__host__ __device__ inline uint32_t record_slot(const RecordLayout& L, uint32_t slot) { return L.slot_map ? L.slot_map[slot] : slot; }
constexpr uint32_t kMaxRecordStride = 128;
struct PublishArgs {
    const uint32_t* count;  // device draw_count
    const uint32_t* idx;
    const float* model;
    const float* dist;
    const uint8_t* is_visible;  // mirror order
    const uint32_t* orig;       // mirror entry -> pool slot (NULL: the mirror is in slot order)
    uint32_t* host_count;       // pinned host memory from here on (device-accessible: hipHostMalloc)
    uint32_t* host_idx;         // NULL: no records emitted
    float* host_model;
    float* host_dist;
    uint8_t* host_is_visible;   // pool-slot order; NULL: not the main pass
    uint32_t occupancy;
    uint8_t* host_records;      // records in the caller's struct layout instead of host_idx / host_model / host_dist (or NULL)
    RecordLayout layout;
};
constexpr uint32_t kMaxPublishViews = 32;  // views of SEVERAL pools per launch (>= GV_MAX_VIEWS, checked in gv_context.cpp)
struct PublishBatch {
    PublishArgs view[kMaxPublishViews];  // blockIdx.y
};
hipError_t launch_publish(const PublishBatch& batch, uint32_t views, uint32_t occupancy, hipStream_t stream);
// records [0, *count) of a view as an array of the caller's structs (device or device-visible host memory)
hipError_t launch_pack_records(const uint32_t* count, const uint32_t* idx, const float* model, const float* dist, const RecordLayout& layout,
                               uint32_t capacity, uint8_t* dst, hipStream_t stream);
hipError_t launch_unpermute_bytes(const uint8_t* src, const uint32_t* orig, uint32_t count, uint8_t* dst, hipStream_t stream);

}  // namespace gv

#include "gv_sort_kernels.hpp"  // gv_sort.hip's launch interface

namespace gv {
static_assert(kMaxSortViews == kMaxPublishViews, "a tick's sort and publish batches hold the same views");

// derives TransformMirror::active_bits from flags[]
// Byte layout of one TransformComponent inside a raw AoS copy (all offsets within `stride`).
struct AosTransformLayout {
    uint32_t stride, entity, position, scale, rotation, self_active, ancestors_active, model_with_ancestors;
};
// dirty (may be NULL): set to 1 for every entry written (subtree-scoped world-matrix sweep)
hipError_t launch_aos_transforms(const uint8_t* raw, const AosTransformLayout& layout, uint32_t first, uint32_t count,
                                 const uint32_t* xinv, XfAB* ab, float2* c, uint8_t* flags, uint8_t* dirty, hipStream_t stream);
hipError_t launch_pack_active(const uint8_t* flags, uint32_t count, unsigned long long* bits, hipStream_t stream);
// The same for dirty MeshRenderComponents (GV_DIRTY_MESH): raw AoS span -> mesh mirror entries. inv: pool slot -> mirror entry
// (NULL: slot order); e2t: the entity -> transform slot table on the device; xinv: transform slot -> mirror entry (or NULL);
// *demoted |= 1 when a candidate does not pair with its own mirror index.
struct AosMeshLayout {
    uint32_t stride, entity, is_enabled, aabb_min, aabb_max;
};
hipError_t launch_aos_meshes(const uint8_t* raw, const AosMeshLayout& layout, uint32_t first, uint32_t count, const uint32_t* inv,
                             const uint32_t* e2t, uint32_t entity_capacity, uint32_t xf_occupancy, const uint32_t* xinv, float4* a, float2* b,
                             uint32_t* link, uint32_t* demoted, hipStream_t stream);
// dirty-range upload into a permuted mirror: dst[idx[k]] = src[k], element size 1, 4, 8 or 16 bytes
hipError_t launch_scatter(const uint32_t* idx, uint32_t count, const void* src, void* dst, uint32_t elem_bytes,
                          hipStream_t stream);
// out[k] = world[xinv[first + k]] (3 float4 per slot)
hipError_t launch_gather_world(const float4* world, const uint32_t* xinv, uint32_t first, uint32_t count, float4* out,
                               hipStream_t stream);

// world matrices (camera = 0) of every transform slot: 3 float4 per slot (float4x3 order)
hipError_t launch_sweep_valu(const TransformMirror& xf, float4* world, hipStream_t stream);
hipError_t launch_sweep_mfma(const TransformMirror& xf, float4* world, hipStream_t stream);
// only the entries whose chain contains an entry flagged in dirty[] (1 byte per mirror entry); same bits
hipError_t launch_sweep_subtree(const TransformMirror& xf, const uint8_t* dirty, float4* world, hipStream_t stream);
// dirty[idx[k]] = 1 for k < count (entries re-mirrored through the scattered dirty-range path)
hipError_t launch_mark_bytes(const uint32_t* idx, uint32_t count, uint8_t* dst, hipStream_t stream);
// Sweep (MFMA or VALU chain) + cull of an exactly paired pool (mesh.mapping == kMapExact) in one pass: world matrices AND the cull
// outputs of one view; same bits as launch_sweep_* followed by launch_cull.
hipError_t launch_sweep_cull(const MeshMirror& mesh, const TransformMirror& xf, const HizDevice& hiz,
                             const ViewParams& vp, const ViewBuffers& out, float4* world, bool mfma, hipStream_t stream);

// Hi-Z pyramid. Level k >= 1 lives at mips + mip_offset[k]; level 0 is the depth image.
// Generic one-level reduction (any size, shaders/hiz.frag:27-56 incl. the odd-size branches).
// rg16f: src_pairs / dst (and HizFusedDst::level) point to packed binary16 pairs, 4 bytes per texel
hipError_t launch_hiz_level(const float* src_depth, const float2* src_pairs, float2* dst, uint32_t sw, uint32_t sh,
                            uint32_t dw, uint32_t dh, uint32_t rule, bool rg16f, hipStream_t stream);
// Levels [first, first + count) by ONE workgroup, one after the other (any sizes; a level's source is the level before it, the
// depth image for level 1). For the small levels of a pyramid: at most kHizTailTexels texels in the first of them.
constexpr uint32_t kHizTailTexels = 8192;  // two levels of this size fit the LDS
struct HizTailArgs {
    const float* depth;
    float2* mips;          // level k >= 1 at mips + offset[k] (texels; packed binary16 pairs when rg16f)
    uint64_t offset[16];
    uint32_t w[16], h[16];
    uint32_t first, count, rule;
};
hipError_t launch_hiz_tail(const HizTailArgs& args, bool rg16f, hipStream_t stream);
// Fused 6-level reduction of 64x64 source tiles through LDS; needs sw % 64 == 0 && sh % 64 == 0.
// dst[l] = level (src+1+l), l = 0..5.
struct HizFusedDst {
    float2* level[6];
};
hipError_t launch_hiz_fused(const float* src_depth, const float2* src_pairs, const HizFusedDst& dst, uint32_t sw,
                            uint32_t sh, bool rg16f, hipStream_t stream);

}  // namespace gv
