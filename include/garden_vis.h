/*
 * garden_vis.h — C-ABI of libgarden_vis.so: the MI355X (gfx950) visibility pass that stands in for
 * Garden's CPU MeshRenderSystem::prepareMeshes + Vulkan HizRenderSystem::downsampleHiz pair.
 *
 * Citations are relative to the reference checkout (cfnptr/garden). The reference has no FFI for
 * this path — it is C++ calling C++ inside one process — so each entry point below names the
 * C++ interface it replaces; INTEGRATION.md shows the shim (an ecsm System) a maintainer adds.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 (GV_OK) or a negative
 * GvStatus; no exceptions or abort() cross this boundary (the reference throws GardenError,
 * include/garden/error.hpp:32-55 — the C++ shim converts). Not re-entrant per context; call from
 * the thread that runs Manager::update() (source/system/input.cpp:361-379). Matrices are
 * column-major float[16] (c0..c3), quaternions xyzw, as in include/garden/system/physics-impl.hpp:45-63.
 *
 * Streams: all device work is enqueued on the context's own stream (gv_stream; created hipStreamNonBlocking: it is NOT
 * ordered against the null stream or any stream of the caller's). Device memory the caller hands in — a GV_MEM_DEVICE depth
 * image, the destination of gv_results_copy_*_device / gv_exchange_shards / gv_exchange_masks — must be ready (its last
 * writer finished, e.g. a fill on another stream) when the call is made, or the caller orders gv_stream() behind its producer
 * itself (hipStreamWaitEvent(gv_stream(ctx), event)); consumers of device results order themselves behind gv_stream().
 */
#ifndef GARDEN_VIS_H
#define GARDEN_VIS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GV_ABI_VERSION 4u /* 3: every exchanged frame is complete (gv_exchange_acquire fills the frame; GV_EXCHANGE_EXACT and
                             gv_exchange_counts are gone); one thread can drive N contexts (gv_exchange_*_all); GV_E_TIMEOUT
                             4: ONE exchange per frame for all its (pool, view) lists (gv_exchange_views[_all]); a rank's results in
                             the WORLD's slots (gv_pool_set_result_mapping, gv_pool_update_index_map); GvStats grew; GvExchangeFrame grew;
                             gv_exchange_init_peers / GV_EXCHANGE_PEER (one process, no communicator) */
#define GV_NONE 0xFFFFFFFFu
#define GV_MAX_POOLS 16u
#define GV_MAX_VIEWS 8u
#define GV_MAX_MIPS 16u

typedef enum GvStatus {
    GV_OK = 0,
    GV_E_ARG = -1,     /* bad argument / unbound pool / capacity exceeded */
    GV_E_HIP = -2,     /* a HIP runtime call failed (text in gv_last_error) */
    GV_E_OOM = -3,     /* host or device allocation failed */
    GV_E_RCCL = -4,    /* collective failed (multi-GPU exchange) */
    GV_E_STATE = -5,   /* call out of order (e.g. cull before bind, Hi-Z query before build) */
    GV_E_NODEVICE = -6, /* no gfx950 device / kernels not loadable: there is NO CPU fallback */
    GV_E_TIMEOUT = -7   /* a bounded wait ran out (a peer rank of the exchange stalled or left): gv_exchange_set_timeout */
} GvStatus;

typedef struct GvCtx GvCtx;

typedef struct GvConfig {
    uint32_t struct_size; /* = sizeof(GvConfig) */
    int32_t device;       /* HIP device ordinal: one context per GPU; the contexts of a node's GPUs may live in one process and be
                             driven by one thread (gv_exchange_*_all) */
    uint32_t hiz_rule;    /* GvHizRule */
    uint32_t flags;       /* GvConfigFlags */
} GvConfig;

typedef enum GvHizRule {
    GV_HIZ_RULE_REFERENCE = 0,   /* shaders/hiz.frag:49-55 exactly as written (odd-height row uses gather .y/.z) */
    GV_HIZ_RULE_CONSERVATIVE = 1 /* full extra row: min/max bound every covered texel */
} GvHizRule;

typedef enum GvConfigFlags {
    GV_CONFIG_PROFILE_EVENTS = 1u << 0,   /* bracket every kernel with hipEvents; durations via gv_stats */
    GV_CONFIG_PROFILE_CULL_ONLY = 1u << 1, /* with PROFILE_EVENTS: only GV_K_CULL is bracketed (2 events per view
                                             instead of ~10 per frame: each event record costs ~2 us of stream time) */
    GV_CONFIG_KEEP_SLOT_ORDER = 1u << 2,  /* keep the device mirror in pool-slot order. Default: entries are ordered
                                             spatially (Morton code of the root ancestor's position) at every full
                                             rebuild so that neighbouring lanes touch neighbouring Hi-Z texels; all
                                             outputs are reported in pool slots either way */
    GV_CONFIG_BLOCK_BOUNDS = 1u << 3,     /* keep a world-space box per 256-entry cull workgroup (built on the device
                                             while the pool's mirror is clean) and let a workgroup whose box lies behind
                                             a frustum plane by more than the rounding margin — or, for a Hi-Z view over a
                                             nested pyramid, wholly behind what the pyramid holds over the box's footprint —
                                             skip its streams. Same results bit for bit (both tests are conservative w.r.t. the
                                             per-entity ones); pools that change every frame are culled without boxes. Pays
                                             off with the default spatial mirror order. Batched views skip a workgroup when
                                             every view does. DEFAULT for pools of more than 262144 slots (round 3); this flag
                                             forces it for pools of every size (they then take neither the one-launch cull + emit
                                             nor the batched tick) */
    GV_CONFIG_LINEAR_SCAN = 1u << 5,      /* never use block bounds: every workgroup reads its streams (the flat loop of
                                             mesh.cpp:137-175, which SURVEY.md §8d prices; bench.py's headline and roofline kernel) */
    GV_CONFIG_HIZ_RG16F = 1u << 4         /* keep the pyramid in the reference's image format (HizRenderSystem::bufferFormat =
                                             SfloatR16G16, render/hiz.hpp:41): levels >= 1 are binary16 (min, max) pairs, half the
                                             bytes. The reference lets the render target round to nearest, which can move a min
                                             up or a max down; here min is rounded toward -inf and max toward +inf, so a texel
                                             still bounds every depth it covers and the occlusion query stays conservative (it may
                                             keep a few boxes an fp32 pyramid would cull, never the other way). Level 0 stays the
                                             fp32 depth image. gv_hiz_read_level returns the stored halfs widened to float. */
} GvConfigFlags;

/* ---- pool binding: replaces LinearPool::getData/getOccupancy (docs/ECS/Components.md:137-170) as
 * consumed at source/system/render/mesh.cpp:119-120,139,360,411,460 ---- */

/* Field byte offsets inside one TransformComponent (include/garden/system/transform.hpp:31-61). */
typedef struct GvTransformLayout {
    uint32_t entity;               /* Component::entity, u32, 0 = free slot */
    uint32_t parent;               /* ID<Entity> parent, u32, 0 = none */
    uint32_t position;             /* f32x4 posChildCount (xyz used) */
    uint32_t scale;                /* f32x4 scaleChildCap (xyz used) */
    uint32_t rotation;             /* quat xyzw */
    uint32_t self_active;          /* volatile bool */
    uint32_t ancestors_active;     /* volatile bool */
    uint32_t model_with_ancestors; /* volatile bool */
} GvTransformLayout;

/* Field byte offsets inside one MeshRenderComponent-derived struct (render/mesh.hpp:45-55). */
typedef struct GvMeshLayout {
    uint32_t entity;
    uint32_t is_enabled;
    uint32_t is_visible; /* written back by gv_results_fetch on main-pass views */
    uint32_t aabb_min;   /* f32x4 */
    uint32_t aabb_max;   /* f32x4 */
} GvMeshLayout;

/* Binds the TransformComponent pool and the entity -> transform-slot map that stands in for
 * Manager::tryGet<TransformComponent>(entity) (mesh.cpp:149) / Manager::get (transform.hpp:206).
 * Re-issue whenever getData() may have moved (create() can reallocate: docs/ECS/Entities.md:44-46).
 * Pointers must stay valid until the next gv_sync()/gv_cull() returns; nothing is retained after.
 * Occupancy may grow from one bind to the next (entities were created): the new slots are appended to the device
 * mirror at the next sync — no rebuild; a smaller occupancy rebuilds it. The same holds for gv_pool_bind. Slots whose
 * contents changed (filled, emptied, edited) are reported with gv_mark_dirty as always. */
int gv_transform_bind(GvCtx* ctx, const void* base, size_t stride, uint32_t occupancy,
                      const GvTransformLayout* layout, const uint32_t* entity_to_transform,
                      uint32_t entity_capacity);

/* Binds one IMeshRenderSystem's component pool (render/mesh.hpp:127-131). pool_id < GV_MAX_POOLS. */
int gv_pool_bind(GvCtx* ctx, uint32_t pool_id, void* base, size_t stride, uint32_t occupancy,
                 const GvMeshLayout* layout);

/* Column (SoA) form of the two binds, for engines and loaders whose components are not an array of structs: every
 * field is its own array, element i at data + i * stride (stride >= the field's width; an AoS pool is the special case
 * data = base + offsetof(field), stride = sizeof(component), which is what gv_transform_bind / gv_pool_bind pass).
 * Same lifetime and dirty-range rules as the AoS binds. position/scale: 3 floats, rotation: 4 floats (xyzw),
 * aabb_min/aabb_max: 3 floats, flags: 1 byte, entity/parent: uint32 entity ids. */
typedef struct GvColumn {
    const void* data;
    uint32_t stride; /* bytes between consecutive elements */
} GvColumn;
typedef struct GvTransformColumns {
    GvColumn entity, parent, position, scale, rotation, self_active, ancestors_active, model_with_ancestors;
} GvTransformColumns;
typedef struct GvMeshColumns {
    GvColumn entity, is_enabled, aabb_min, aabb_max;
    void* is_visible;           /* write-back target of gv_results_fetch (may be NULL) */
    uint32_t is_visible_stride; /* bytes between consecutive isVisible bytes */
} GvMeshColumns;
int gv_transform_bind_columns(GvCtx* ctx, const GvTransformColumns* columns, uint32_t occupancy,
                              const uint32_t* entity_to_transform, uint32_t entity_capacity);
int gv_pool_bind_columns(GvCtx* ctx, uint32_t pool_id, const GvMeshColumns* columns, uint32_t occupancy);
/* Optional per-slot READY COUNT of a pool: what a derived system's getReadyMeshesAsync returns for a mesh that passed
 * the frustum test — 0 while its resources are not loaded (SpriteRenderSystem: descriptorSet, sprite.cpp:90-97;
 * UiLabelSystem: text data ready, label.cpp:271-276), otherwise the number of instances it draws (1 for the default
 * predicate, render/mesh.hpp:142-146). Element i at data + i * stride, `width` = 1 (uint8) or 4 (uint32) bytes.
 * A slot whose count is 0 is treated exactly like the reference treats readyCount == 0 (mesh.cpp:158-165): not drawn,
 * isVisible = false; GvResult.instance_count becomes the sum of the counts of the drawn meshes (mesh.cpp:174).
 * Same lifetime rules as gv_pool_bind (re-issue when the storage may have moved); changed counts are reported with
 * gv_mark_dirty(GV_DIRTY_MESH). data == NULL removes the column (every slot counts 1 again).
 * Which meshes are DRAWN is decided by the counts as mirrored at the cull; instance_count and the instance bases are summed on the
 * host from the column as it stands when the results are fetched (the reference reads getInstancesAsync at draw time too,
 * mesh.cpp:596): change counts between frames — after a frame's results have been read, before the next gv_cull — not in between. */
int gv_pool_bind_ready(GvCtx* ctx, uint32_t pool_id, const void* data, uint32_t stride, uint32_t width);

typedef enum GvDirtyKind {
    GV_DIRTY_TRANSFORM = 0, /* TRS / active flags of transform slots [first, first+count) changed
                               (setPosition/Scale/Rotation transform.hpp:74-104, setActive transform.cpp:75-127) */
    GV_DIRTY_HIERARCHY = 1, /* count > 0: transform slots [first, first+count) were re-parented (setParent
                               transform.cpp:130-195): their links are re-gathered in place and depth / cycles
                               re-validated; count == 0: entities came or went (destroy :30-73, create): rebuild and
                               re-order the whole mirror */
    GV_DIRTY_MESH = 2       /* mesh slots changed; pool id in the top 4 bits of `first` */
} GvDirtyKind;
int gv_mark_dirty(GvCtx* ctx, uint32_t kind, uint32_t first, uint32_t count);

/* Re-derives the whole device mirror (parent slots, transform slot per mesh, flags) from the bound
 * pools and uploads it. Implied by the first gv_sync after a bind. */
int gv_hierarchy_rebuild(GvCtx* ctx);
/* Uploads whatever is dirty (host gather AoS -> SoA staging, then H2D). Called by gv_cull too. */
int gv_sync(GvCtx* ctx);

/* ---- per-frame inputs: replaces the CommonConstants read at mesh.cpp:866-869,899-902 and the
 * per-cascade viewProj/cameraOffset of renderShadows (mesh.cpp:795-847) ---- */
typedef struct GvView {
    float view_proj[16];      /* cc.viewProj = projection * view, view translation zeroed (graphics.cpp:201,243) */
    float camera_position[4]; /* cc.cameraPos (graphics.cpp:202); xyz used */
    float camera_offset[4];   /* shadow cascades (render/mesh.hpp:166); zero for the main camera */
    int8_t shadow_pass;       /* < 0: main pass, isVisible is produced (mesh.cpp:121) */
    uint8_t use_hiz;          /* also run the Hi-Z occlusion query (needs gv_hiz_build first) */
    uint8_t distance_2d;      /* sorted twin key translation.z + 1 (mesh.cpp:250) */
    uint8_t emit_records;     /* 1: produce visibleIdx/bakedModel/distanceSq; 0: isVisible + count only */
} GvView;

/* Host-visible results of one view: the SoA form of UnsortedBuffer::combinedMeshes[0..drawCount)
 * (render/mesh.hpp:191-217). Pointers are library-owned pinned memory, valid until the next
 * gv_cull on this context. Record order is deterministic (reproducible run to run): ascending pool slot with
 * GV_CONFIG_KEEP_SLOT_ORDER, otherwise the mirror's spatial order; gv_sort orders them by distance. The
 * reference's order is fetch_add arrival order (mesh.cpp:177), i.e. unspecified, and its consumers sort by
 * distanceSq afterwards (mesh.cpp:548-551). is_visible is always indexed by pool slot. */
typedef struct GvResult {
    const uint32_t* visible_idx; /* pool slot; componentOffset = visible_idx * stride (mesh.cpp:170) */
    const float* baked_model;    /* 12 floats per record: c0.xyz c1.xyz c2.xyz c3.xyz (float4x3, mesh.cpp:171) */
    const float* distance_sq;    /* mesh.cpp:172 / :250-251 */
    const uint8_t* is_visible;   /* one byte per pool slot; NULL for shadow passes */
    uint32_t draw_count;         /* UnsortedBuffer::drawCount (render/mesh.hpp:210) */
    uint32_t instance_count;     /* sum of the drawn meshes' ready counts (mesh.cpp:174): == draw_count unless a ready column
                                    with counts above 1 is bound (gv_pool_bind_ready) */
} GvResult;

/* Device-side cull of one pool against `view_count` views: frustum test (+ Hi-Z query) + compaction.
 * Asynchronous: returns once the work is enqueued on the context's stream. Replaces
 * MeshRenderSystem::prepareMeshes' threaded loop (mesh.cpp:331-553 -> :111-184).
 * Everything a view's results consist of is enqueued by this call, in stream order: a consumer on gv_stream() may use the
 * pointers of gv_results_device without calling into the library again (round 3 held a single occlusion view's record emission
 * back until the next gv_hiz_build; withdrawn in round 4 — only a loop that never looks at its results gained from it). */
int gv_cull(GvCtx* ctx, uint32_t pool_id, const GvView* views, uint32_t view_count);
/* Blocks until all enqueued work is done. */
int gv_wait(GvCtx* ctx);
/* Waits, copies view `view_index`'s records to pinned host memory, and — for a main-pass view — when
 * write_back != 0 scatters isVisible into the bound pool (mesh.cpp:144,152,161,166). */
int gv_results_fetch(GvCtx* ctx, uint32_t view_index, int write_back, GvResult* out);
/* draw count only (4-byte readback). */
int gv_result_count(GvCtx* ctx, uint32_t view_index, uint32_t* draw_count);
/* Device pointers of view `view_index`'s compacted list, for an exchange step that stays on the GPU
 * (multi-GPU all-gatherv). Valid until the next gv_cull. */
typedef struct GvDeviceResult {
    const void* visible_idx; /* uint32_t[draw_count], device memory */
    const void* baked_model; /* float[12 * draw_count] */
    const void* distance_sq; /* float[draw_count] */
    const void* is_visible;  /* uint8_t[occupancy] or NULL; indexed by MIRROR entry (pool slot only with
                                GV_CONFIG_KEEP_SLOT_ORDER) — use gv_results_fetch for the pool-slot view */
    const void* draw_count;  /* uint32_t, device memory */
} GvDeviceResult;
int gv_results_device(GvCtx* ctx, uint32_t view_index, GvDeviceResult* out);
/* Copies visible_idx (+ `index_base` added to each) into caller-owned device memory. */
int gv_results_copy_idx_device(GvCtx* ctx, uint32_t view_index, void* dst_device, uint32_t capacity,
                               uint32_t index_base);
/* Exchange shard for the multi-GPU all-gather: dst[0] = draw_count (the true count, also when it
 * exceeds `capacity`), dst[1 .. 1 + min(draw_count, capacity)) = visible_idx + index_base. dst holds 1 + capacity
 * uint32. No host synchronisation: ranks can gather fixed-size padded shards and read the counts from the headers. */
int gv_results_copy_shard_device(GvCtx* ctx, uint32_t view_index, void* dst_device, uint32_t capacity,
                                 uint32_t index_base);

/* The same visible set as a BIT per entry of the pool's device mirror: dst[0] = draw_count, bit (e & 31) of dst[1 + (e >> 5)]
 * = mirror entry e is in the view's visible list; dst holds 1 + word_count uint32, word_count >= ceil(occupancy / 32). The
 * size does not depend on the view: 1/32 of a word per slot where the index list costs a word per visible slot — smaller above
 * ~3 % visibility, 1/7 of the list at the bench's 21 % — and an equal-size all-gather by construction: the encoding for dense
 * views on the links of a multi-GPU node. It is the cull kernel's own output, so producing it is a 1.6 MB copy per 12.5 M
 * entries (bits in pool-slot order would cost a scatter per frame: measured, +126 us). Entry e is pool slot
 * gv_pool_mirror_slots()[e]: that table only changes when the mirror is rebuilt, grown or re-ordered (gv_hierarchy_rebuild, a pool
 * bound with another occupancy; gv_pool_mirror_epoch says when), so consumers fetch it once per change — with GV_CONFIG_KEEP_SLOT_ORDER it is the identity. Main-pass
 * views of any pool, and every view that took the ordinary cull launch (GV_E_STATE otherwise). No host synchronisation. */
int gv_results_copy_mask_device(GvCtx* ctx, uint32_t view_index, void* dst_device, uint32_t word_count);
/* entry_to_slot[e] = pool slot of mirror entry e, e < min(occupancy, capacity); host memory. Synchronises the mirror first. */
int gv_pool_mirror_slots(GvCtx* ctx, uint32_t pool_id, uint32_t* entry_to_slot, uint32_t capacity);
/* A counter that changes whenever that table does — a (re)build, slots appended after a bind with a larger occupancy, the
 * re-order of the mirror after entity churn: a consumer of the bit shards compares it with the value it fetched the table at.
 * Synchronises the mirror first (like gv_pool_mirror_slots: the two are consistent with each other). */
int gv_pool_mirror_epoch(GvCtx* ctx, uint32_t pool_id, uint64_t* epoch);

/* A spatial tile's pool slots are not a contiguous range of the world's (roots -> tile, descendants
 * follow): `global_ids[slot]` is the id the exchange should carry for pool slot `slot` (e.g. the mesh slot in the
 * unpartitioned world, garden_amd/multi.py::partition_world's mesh_global). Uploaded once; from then on
 * gv_results_copy_idx_device / gv_results_copy_shard_device / gv_exchange_shards write global_ids[visible_idx] +
 * index_base instead of visible_idx + index_base whenever the table covers the culled pool. count == 0 removes it. */
int gv_pool_set_index_map(GvCtx* ctx, uint32_t pool_id, const uint32_t* global_ids, uint32_t count);
/* Entries [first, first + count) of that table replaced (a few slots of a rank's share changed hands: a root crossed into another
 * rank's cell and its tree moved, SURVEY.md §8e "re-bin only roots whose position crosses a cell"); first + count may exceed the
 * table's size — it grows (slots appended to the share). GV_NONE marks a slot that stands for no world slot (a hole in the share:
 * never visible, skipped by the write-back below). Ordered on gv_stream(ctx) behind the work already queued. */
int gv_pool_update_index_map(GvCtx* ctx, uint32_t pool_id, uint32_t first, const uint32_t* global_ids, uint32_t count);
/* A rank's results delivered in the WORLD's numbering (one process, N contexts: the host fills the engine's buffers from every
 * rank's results, mesh.cpp:144-183). flags:
 *   GV_RESULTS_MAP_RECORDS  records in the pool's record layout (gv_pool_set_record_layout) carry componentOffset =
 *                           index_map[slot] * component_stride (mesh.cpp:170 in the engine's own pool): a rank's records are copied
 *                           into combinedMeshes as they are
 *   GV_RESULTS_MAP_VISIBLE  the write-back of gv_pool_results_fetch(write_back != 0) stores slot i's isVisible byte at
 *                           visible_base + index_map[i] * visible_stride — straight into the engine's pool (mesh.cpp:144,152,161,
 *                           166), wherever the rank's share keeps slot i — instead of into the bound share; entries that are GV_NONE
 *                           or >= visible_count are skipped. The caller guarantees [visible_base, + visible_count * visible_stride)
 *                           until the fetch returns; ranks write disjoint slots.
 * Both need the pool's index map (GV_E_STATE at the fetch while it does not cover the culled pool). GvResult.is_visible stays in
 * the share's own slots. Not available together with gv_pool_bind_ready (GV_E_STATE). flags == 0 removes the mapping. */
#define GV_RESULTS_MAP_RECORDS 1u
#define GV_RESULTS_MAP_VISIBLE 2u
int gv_pool_set_result_mapping(GvCtx* ctx, uint32_t pool_id, uint32_t flags, void* visible_base, size_t visible_stride, uint32_t visible_count);

/* Results are kept per (pool, view). The view-indexed calls above address the pool of the most recent gv_cull; these
 * name the pool, so that a frame can issue the culls (and sort requests) of ALL its mesh systems first and read the
 * results afterwards — the reference's prepareMeshes does the same with its thread pool (dispatch every system's tasks,
 * mesh.cpp:408-546, then one wait, :548) — the device works through the systems back to back, and for engine-sized
 * pools (up to 262144 slots) the first fetch publishes the results of every pool with ONE launch and ONE
 * synchronisation; the other fetches find theirs in the pinned host buffers. A (pool, view)'s results stay valid until
 * the next gv_cull of that pool. */
/* Records in the ENGINE'S OWN struct layout. The reference appends `UnsortedMesh { psize componentOffset; float4x3
 * bakedModel; float distanceSq; }` / `SortedMesh { ... uint32 bufferIndex; }` (render/mesh.hpp:191-205) to combinedMeshes;
 * their padding depends on the math library's alignment of float4x3, so the layout is described, not assumed: once a
 * pool has a record layout, the results of its views are ALSO delivered as an array of such structs in pinned host memory
 * (gv_pool_results_records) — combinedMeshes is then filled with one memcpy instead of a loop over three arrays, and the
 * SoA pointers of GvResult (visible_idx / baked_model / distance_sq) are NULL for that pool (draw_count, instance_count and
 * is_visible are delivered as always). stride: bytes per record, a multiple of 16, at most 128; offsets inside the record:
 * component_offset (uint64 = visible slot * component_stride, mesh.cpp:170), baked_model (12 floats, mesh.cpp:171),
 * distance_sq (float, mesh.cpp:172 / :250-251), buffer_index (uint32 = buffer_index_value, mesh.cpp:252; GV_NONE: the struct
 * has none); bytes not covered by a field are zero. layout == NULL removes it. */
typedef struct GvRecordLayout {
    uint32_t stride;
    uint32_t component_offset, baked_model, distance_sq, buffer_index;
    uint32_t component_stride;   /* getMeshComponentSize() */
    uint32_t buffer_index_value; /* SortedMesh::bufferIndex of this pool's system */
} GvRecordLayout;
int gv_pool_set_record_layout(GvCtx* ctx, uint32_t pool_id, const GvRecordLayout* layout);
/* After gv_pool_results_fetch of the same (pool, view): the records [0, *count) in the pool's record layout (library-owned
 * pinned memory, valid until the next gv_cull of that pool). GV_E_STATE when the pool has no record layout. */
int gv_pool_results_records(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, const void** records, uint32_t* count);
/* The records of (pool, view) into the CALLER'S array — `UnsortedBuffer::combinedMeshes.data()`, which the reference grows and
 * never shrinks (mesh.cpp:377-395): the fetch leaves each frame's records [0, draw_count) there (one memcpy from the library's
 * pinned buffer, inside the fetch — the engine's own copy loop disappears) and gv_pool_results_records returns `records` itself.
 * The array is never page-locked: letting the device write it in place (round 2: hipHostRegister once per address, 4 us less
 * per 10 k-entity tick) made LATER, unrelated copies into pageable memory abort inside the runtime now and then once such an
 * array had been freed and its addresses reused (round 3: 4 of 16 runs of the GPU test tier). The pool needs a
 * record layout; `bytes` >= occupancy * stride of every pool culled while the target is set (GV_E_ARG at the fetch otherwise);
 * `records` 16-byte aligned. The range must stay allocated until it is replaced (another call for the same pool and view),
 * removed (records == NULL) or the context is destroyed. A range that is found unmapped when it is WRITTEN fails that fetch with
 * GV_E_STATE; one found unmapped when it is let go is counted in GvStats::record_targets_lost and described in gv_last_error —
 * the call that replaces it succeeds, the new target is in place (a freed heap block that is still mapped cannot be told from a
 * live one). */
int gv_pool_set_record_target(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, void* records, size_t bytes);

/* The first instance index of every fetched record: bases[k] = sum of the ready counts (gv_pool_bind_ready; 1 per record
 * without) of records [0, k), bases[*count] = the view's instance_count. This is what `instanceCount.fetch_add(
 * getInstancesAsync(view))` hands out draw by draw in the reference's render loops (mesh.cpp:596-599, :624-627, :709-712),
 * available up front and in record order (after gv_sort: in sorted order), so a threaded draw loop needs no atomic and
 * gives the same instance layout every frame. Library-owned host memory ([*count + 1] words), valid until the next gv_cull
 * of the pool; fetches the results first if that has not happened yet. GV_E_STATE for a count-only view. */
int gv_pool_results_instance_bases(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, const uint32_t** bases, uint32_t* count);

/* A tick of engine-sized pools (the reference's everyday 10^3..10^4 entities per mesh system) is bound by launches,
 * not by bytes. Between gv_cull_batch_begin and the first call that reads results (gv_pool_results_* / gv_results_* /
 * gv_wait, or gv_cull_batch_end), gv_cull of a pool of up to 32768 slots whose views all emit records only RECORDS the
 * cull; the first read then launches all recorded culls as ONE kernel, their emits as ONE kernel, the requested sorts
 * (gv_pool_sort; pools of up to 16384 slots) as ONE kernel and publishes every view's results to the host with ONE kernel and ONE synchronisation —
 * four launches per frame however many mesh systems and shadow passes there are. Other culls (larger pools, count-only
 * views, GV_SWEEP_WITH_CULL, block bounds) run at once as usual. Same results. A bind or gv_mark_dirty of a pool a recorded cull
 * reads (its mesh pool; the transform pool, gv_sync and gv_hierarchy_rebuild concern every recorded cull) first launches what
 * has been recorded so far — the recorded culls see the pools as they were when gv_cull was called — and recording goes on. */
int gv_cull_batch_begin(GvCtx* ctx);
int gv_cull_batch_end(GvCtx* ctx);
int gv_pool_results_fetch(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, int write_back, GvResult* out);
int gv_pool_result_count(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, uint32_t* draw_count);
int gv_pool_results_device(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, GvDeviceResult* out);
int gv_pool_sort(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, int descending);

/* ---- multi-GPU exchange (one process per GPU): the all-gatherv of the compacted visible lists over RCCL.
 * Replaces what the reference does inside one address space — every worker appends its range's records to the shared
 * array with `drawCount.fetch_add` + memcpy into combinedMeshes (source/system/render/mesh.cpp:177-183).
 * Rank 0 calls gv_exchange_unique_id and hands the 128 bytes to the other ranks by its own means (the engine's IPC, a
 * file, MPI ...); every rank then calls gv_exchange_init with its own context. RCCL is dlopen'ed at the first call — the
 * copy already in the process if there is one, else librccl.so.1, or the library the environment variable GV_RCCL_LIBRARY
 * names; failures return GV_E_RCCL with the RCCL text in gv_last_error. All ranks make the same calls in the same order. */
#define GV_EXCHANGE_ID_BYTES 128
#define GV_EXCHANGE_MAX_RANKS 64u
int gv_exchange_unique_id(void* out_id_128_bytes);
int gv_exchange_init(GvCtx* ctx, const void* unique_id_128_bytes, int rank, int world_size);

/* One host thread, N GPUs (the reference is ONE process with ONE Manager, source/editor/entry.cpp:135): the *_all forms take the
 * N contexts of the node's GPUs (contexts[r] = rank r) and issue their collectives inside ONE ncclGroupStart / ncclGroupEnd, which is
 * what RCCL requires of a thread that drives several devices. gv_exchange_init_all makes the unique id itself. A communicator
 * started with one form is used through that form only (per-rank calls from N threads or processes, or the *_all calls from one). */
int gv_exchange_init_all(GvCtx* const* contexts, int world_size);

/* One process, N GPUs, NO communicator: the devices of a node reach each other over xGMI with plain stores, so the ranks one thread
 * drives need neither RCCL nor a size prediction. gv_exchange_init_peers checks and enables peer access between every pair of the
 * contexts' devices (contexts may share a device) and fixes the travel pattern GV_EXCHANGE_PEER: per frame ONE kernel per rank
 * writes the words its list really has — known on the device, never on the host — from its staging shard straight into its row of
 * EVERY rank's rows (wide stores, all links at once), ordered between the ranks by events alone. Rows have room for a rank's whole
 * pool, so no frame is ever cut and nothing travels twice: tail_words / cut_ranks stay 0, room[r] = the row's capacity in entries,
 * travelled_words[r] (acquired frames) = 1 + counts[r]. Everything else — gv_exchange_visible_all / gv_exchange_views_all /
 * gv_exchange_acquire_all, the headers on the host, the bounded waits, alternating buffers — is as for the other patterns.
 * GV_E_STATE: some device cannot reach another (use gv_exchange_init_all). gv_exchange_shutdown / gv_destroy of ONE member drains
 * every member's exchange stream and dissolves the group (the others return GV_E_STATE until they are initialised again). */
int gv_exchange_init_peers(GvCtx* const* contexts, int world_size);

/* The per-frame exchange — the gather of mesh.cpp:177-183, which never loses a record: EVERY frame a consumer can acquire holds
 * every rank's complete list.
 *
 * gv_exchange_visible enqueues, on the context's stream, this rank's WHOLE shard [draw_count, visible_idx + index_base ...] into a
 * library-owned staging buffer (gv_results_copy_shard_device; the pool's gv_pool_set_index_map table applies) — the last thing
 * gv_stream(ctx) does for the frame, so the caller's next gv_hiz_build / gv_cull run while this list is still on the links — and,
 * on a second stream of the library's, the gather into library-owned rows (row r = rank r's shard). How many words of each rank's
 * row travel is PREDICTED from the previous frame's row headers, which every rank holds (they reach the host through pinned memory
 * behind that frame's collective): rank r gets count + max(count / 8, 1024) entries of room, rounded up to 1024 — grown at once,
 * given back when the list has shrunk by a quarter. Every rank sees the same headers, so every rank derives the same sizes.
 *
 * A prediction can be short (a camera cut). The frame is SETTLED — by gv_exchange_acquire of that frame or by the next
 * gv_exchange_visible, whichever comes first, and before any later collective of the communicator on every rank — by reading its
 * own headers on the host: where a header exceeds the room its row had, the missing tails travel in a second, exactly sized
 * exchange (one grouped ncclBroadcast per short rank, out of the staging buffer that still holds the whole list) to their place in
 * the rows, and only then is the frame handed out. out->cut_ranks is a statistic, not a caveat.
 *
 * Host waits: settling polls a pinned word that the frame's own collective writes — bounded (gv_exchange_set_timeout, default
 * 30 s; GV_E_TIMEOUT: a peer stalled), never a device-wide synchronisation. A caller that enqueues the next frame's cull before it
 * acquires this frame's rows keeps the device busy throughout. flags must be 0. Buffers alternate: a frame's rows stay valid until
 * the exchange after the next. */
typedef struct GvExchangeFrame {
    const void* gathered_device; /* uint32 [world_size][row_words], device memory owned by the library; NULL until the frame has
                                    been acquired (a second exchange may move the rows) */
    uint32_t row_words;          /* words between two rows, a multiple of 4 (>= 1 + the largest list) */
    uint32_t world_size;
    uint64_t frame;              /* 0, 1, 2 ... since gv_exchange_init */
    uint32_t room[GV_EXCHANGE_MAX_RANKS];            /* list entries rank r's row was PREDICTED to need (what travelled first) */
    uint32_t travelled_words[GV_EXCHANGE_MAX_RANKS]; /* of rank r's row, the words that crossed a link in the first exchange (header
                                                        included; the equal-size all-gather moves whole rows whatever the rooms) */
    uint32_t counts[GV_EXCHANGE_MAX_RANKS];          /* acquired frames: rank r's draw count = the entries behind row r's header */
    uint32_t tail_words[GV_EXCHANGE_MAX_RANKS];      /* acquired frames: words of rank r's list that travelled in the second exchange */
    uint64_t cut_ranks;          /* acquired frames: bit r = rank r's list outgrew its room and was completed (statistic) */
    uint32_t complete;           /* 1: acquired — every row holds its rank's whole list */
    uint32_t mode;               /* GvExchangeMode the rows travelled by */
    void* ready_event;           /* acquired frames: hipEvent_t behind the complete rows (hipStreamWaitEvent on a consumer's stream) */
    /* gv_exchange_views (0 for the single-list forms): the frame carries `items` lists per rank. Row r = [items + total, c_0 ..
     * c_{items-1}, list 0 (c_0 entries), list 1 ...]: the header counts the table too, so counts[] / room[] / tail_words[] above
     * are in words behind the header. item_counts (acquired frames; library-owned host memory, valid as long as the rows):
     * item_counts[r * items + i] = c_i of rank r. */
    uint32_t items;
    const uint32_t* item_counts;
} GvExchangeFrame;
int gv_exchange_visible(GvCtx* ctx, uint32_t view_index, uint32_t index_base, uint32_t flags, GvExchangeFrame* out);
/* view_index / index_base: [world_size] (index_bases NULL: all 0); frames: [world_size] outputs. */
int gv_exchange_visible_all(GvCtx* const* contexts, int world_size, const uint32_t* view_indices, const uint32_t* index_bases,
                            uint32_t flags, GvExchangeFrame* frames);
/* The same for a pool that is named (the view-indexed forms above address the pool of the most recent gv_cull): a frame that culls
 * several mesh systems first — gv_cull_batch_begin — exchanges each system's views afterwards. pool_id is the same on every rank. */
int gv_pool_exchange_visible(GvCtx* ctx, uint32_t pool_id, uint32_t view_index, uint32_t index_base, uint32_t flags, GvExchangeFrame* out);
int gv_pool_exchange_visible_all(GvCtx* const* contexts, int world_size, uint32_t pool_id, const uint32_t* view_indices,
                                 const uint32_t* index_bases, uint32_t flags, GvExchangeFrame* frames);
/* ONE exchange for ALL the lists of a frame — the reference dispatches every mesh system's tasks and waits once
 * (source/system/render/mesh.cpp:408-546, :548): `items` names the (pool, view) pairs of the frame, the same list on every rank; a
 * rank's row is [item_count + total entries, c_0 .. c_{item_count-1}, list 0, list 1 ...] (lists packed back to back, each as
 * gv_pool_exchange_visible would send it: the pool's index map applied, index_base added), sized, completed and acquired as ONE
 * frame exactly like the single-list forms (the room prediction works on the row's words; a cut row's tail is one contiguous piece).
 * item_count <= GV_EXCHANGE_MAX_ITEMS. A consumer finds list i of rank r at row r + 1 + item_count + sum(c_0 .. c_{i-1}).
 * GvStats::exchanges counts frames sent, whatever their form. */
#define GV_EXCHANGE_MAX_ITEMS 128u
typedef struct GvExchangeItem {
    uint32_t pool_id, view_index, index_base;
} GvExchangeItem;
int gv_exchange_views(GvCtx* ctx, const GvExchangeItem* items, uint32_t item_count, uint32_t flags, GvExchangeFrame* out);
int gv_exchange_views_all(GvCtx* const* contexts, int world_size, const GvExchangeItem* items, uint32_t item_count, uint32_t flags,
                          GvExchangeFrame* frames);
/* Settles frame `frame` (one of the last two) if that has not happened yet — waits for its headers on the host, completes cut rows —
 * makes gv_stream(ctx) wait for the complete rows and fills *out (NULL: not wanted). Every rank acquires, or none does: the
 * completing exchange is a collective (ranks that skip it meet it inside their next gv_exchange_visible). */
int gv_exchange_acquire(GvCtx* ctx, uint64_t frame, GvExchangeFrame* out);
int gv_exchange_acquire_all(GvCtx* const* contexts, int world_size, uint64_t frame, GvExchangeFrame* frames);
/* Upper bound of every host wait of the exchange, in milliseconds (0: the default, 30 000; also GV_EXCHANGE_TIMEOUT_MS at
 * gv_exchange_init). A wait that runs out aborts the communicator on the spot (ncclCommAbort: the collective that will never finish
 * must not keep the device — and with it every hipFree / synchronise of the process — waiting) and returns GV_E_TIMEOUT; later
 * exchange calls return GV_E_STATE until gv_exchange_shutdown + gv_exchange_init. */
int gv_exchange_set_timeout(GvCtx* ctx, uint32_t milliseconds);

/* The same exchange with caller-owned buffers and caller-chosen sizes: row r of gathered_device (world_size * (capacity + 1)
 * uint32, device memory) = rank r's shard. capacities == NULL: every row travels whole (capacity + 1 words). Otherwise
 * capacities[world_size], each <= capacity and the same list on every rank: the direct patterns (GV_EXCHANGE_P2P /
 * _BROADCAST) move 1 + capacities[r] words of rank r's row, and this rank's shard is cut to capacities[rank] entries; the
 * equal-size all-gather moves capacity + 1 words per row regardless. No host synchronisation; a header above the row's room =
 * that rank's list was cut — this form is the caller's own sizing, with nothing behind it: use gv_exchange_visible for lists that
 * must arrive whole. */
int gv_exchange_shards(GvCtx* ctx, uint32_t view_index, uint32_t capacity, const uint32_t* capacities, uint32_t index_base,
                       void* gathered_device);
/* The exchange with bit shards (gv_results_copy_mask_device): row r of gathered_device = rank r's [draw_count, one bit per
 * mirror entry] (word_count + 1 uint32 per row; word_count the same on every rank, >= ceil(occupancy / 32) of the largest
 * pool). A fixed size whatever the view — the encoding for dense views (1/32 word per entry). */
int gv_exchange_masks(GvCtx* ctx, uint32_t view_index, uint32_t word_count, void* gathered_device);
/* Drains what is still queued (the same bounded wait: a frame that was sent and never acquired may sit behind a peer that has left —
 * GV_E_TIMEOUT / GV_E_RCCL then, the communicator aborted instead of drained) and releases the communicator, streams and rows in every
 * case; gv_destroy and a second gv_exchange_init do the same. */
int gv_exchange_shutdown(GvCtx* ctx);
/* How the rows travel (same result rows either way). The node's xGMI fabric is point to point and fully connected
 * (SURVEY.md §5, §8e): a ring all-gather serialises world-1 hops, the direct forms use every link at once and move only
 * what each rank's list needs. Default GV_EXCHANGE_ALLGATHER, or the environment variable GV_EXCHANGE_MODE (allgather | p2p |
 * broadcast) read by gv_exchange_init. */
typedef enum GvExchangeMode {
    GV_EXCHANGE_ALLGATHER = 0, /* one equal-size ncclAllGather */
    GV_EXCHANGE_P2P = 1,       /* one ncclGroup of ncclSend/ncclRecv pairs with every peer */
    GV_EXCHANGE_BROADCAST = 2, /* one ncclBroadcast per root, grouped */
    GV_EXCHANGE_PEER = 3       /* gv_exchange_init_peers only: direct stores into every rank's rows, no communicator */
} GvExchangeMode;
int gv_exchange_set_mode(GvCtx* ctx, uint32_t mode);

/* Sorts view `view_index`'s compact records on the device by distanceSq: ascending (descending == 0) as sortMeshes
 * does for unsorted buffers — front to back, operator< at render/mesh.hpp:196 — or descending for the sorted /
 * translucent ones (render/mesh.hpp:204; mesh.cpp:265-328). Stable: equal keys keep the order the records were
 * emitted in (std::sort in the reference leaves ties unspecified).
 * Call after gv_cull (records requested), before gv_results_fetch / gv_results_device. For pools of up to 2^20 slots
 * the launch is deferred to the first call that reads the records (gv_results_*, gv_exchange_*, gv_wait), where the pending
 * sorts of ALL views of ALL pools share their launches: one launch for the views of up to 16384 slots; for the larger ones one
 * set of launches — a one-launch rank sort AND the radix launches, list = blockIdx.y, the record count on the device deciding
 * per list which of them works (short lists: one launch's worth of time). A frame of many mid-sized mesh systems is bound by
 * its launches, not by bytes. Larger pools sort at once. */
int gv_sort(GvCtx* ctx, uint32_t view_index, int descending);

/* ---- scene ingest: a Garden scene file straight into column pools, no component AoS ----
 * Reads what ResourceSystem::loadScene (source/system/resource.cpp:2421-2510) hands to TransformSystem::deserialize /
 * postDeserialize (source/system/transform.cpp:517-583) and to the mesh systems' deserialize ("aabb", "isEnabled",
 * e.g. source/system/render/sprite.cpp:206-207), with the JSON typing rules of source/json-serialize.cpp. Entity ids
 * are 1, 2, ... in file order (a fresh Manager); transform / mesh slots are in order of appearance; parents are
 * resolved by uid after the whole file is read, in file order, with setParent's semantics (ancestorsActive from the
 * parent at that moment, transform.cpp:129-195). Parsing needs no device. */
typedef struct GvScene GvScene;
typedef struct GvScenePool {
    const char* component_type; /* the component's ".type" string, e.g. "Model", "Sprite" */
    uint32_t pool_id;           /* the gv_pool_bind id its meshes go to */
} GvScenePool;
typedef struct GvSceneInfo {
    uint32_t entity_count;     /* entities created (1 .. entity_count) */
    uint32_t transform_count;
    uint32_t mesh_count[GV_MAX_POOLS];
    uint32_t skipped_entities;   /* "components": [] — the reference creates no entity for them */
    uint32_t other_components;   /* components of types this pass does not read */
    uint32_t duplicate_uids;     /* later transforms with a uid already seen (the first keeps it) */
    uint32_t self_parents;       /* parent uid == own uid: no link */
    uint32_t unresolved_parents; /* parent uid not in the file: stays a root */
} GvSceneInfo;
/* flags: GV_SCENE_ADD_ROOT_ENTITY = loadScene's addRootEntity (resource.cpp:2398-2407,2497-2502): entity 1 is a default
 * transform and every loaded transform is parented to it before the file's own parent links are applied.
 * On failure returns GV_E_ARG and writes a message into `error` (may be NULL). */
#define GV_SCENE_ADD_ROOT_ENTITY 1u
int gv_scene_parse_json(const char* text, size_t length, const GvScenePool* pools, uint32_t pool_count, uint32_t flags,
                        GvScene** out_scene, char* error, size_t error_capacity);
/* The same scene as the packed builds carry it: the BSON document nlohmann::json::to_bson makes of the scene's JSON
 * (json2bson.cpp:41-66), which ResourceSystem::loadScene hands to JsonDeserializer::load(vector<uint8>) =
 * json::from_bson (resource.cpp:2359-2377, json-serialize.cpp:338-344). Same results as gv_scene_parse_json on the
 * text the document was made from: from_bson keeps the value categories the readers test (double -> float literal,
 * int32 / int64 / uint64 -> integer, bool, string, document, array by element order). Element types nlohmann's
 * from_bson rejects are rejected (GV_E_ARG). */
int gv_scene_parse_bson(const void* data, size_t length, const GvScenePool* pools, uint32_t pool_count, uint32_t flags,
                        GvScene** out_scene, char* error, size_t error_capacity);
void gv_scene_destroy(GvScene* scene);
int gv_scene_info(const GvScene* scene, GvSceneInfo* out);
/* The scene's columns (owned by the scene, valid until gv_scene_destroy); any out pointer may be NULL. */
int gv_scene_transform_columns(const GvScene* scene, GvTransformColumns* columns, uint32_t* occupancy,
                               const uint32_t** entity_to_transform, uint32_t* entity_capacity, const uint64_t** uids);
int gv_scene_mesh_columns(GvScene* scene, uint32_t pool_id, GvMeshColumns* columns, uint32_t* occupancy);
/* Binds the transform columns and every mapped mesh pool to `ctx` and schedules a full mirror build. The scene must
 * outlive the binding. For a tile (gv_scene_extract_tile) the pools' slot -> world-slot tables are installed too
 * (gv_pool_set_index_map), so exchange shards carry the world's mesh slots. */
int gv_scene_bind(GvCtx* ctx, GvScene* scene);
/* One spatial tile of a scene as a scene of its own (one process per GPU, entities sharded by spatial
 * tile; nearest reference analogue: the contiguous range split of ThreadPool::addItems, source/thread-pool.cpp:173-200).
 * The world cube of edge `side`, centred on the origin, is cut into grid[0] x grid[1] x grid[2] cells; `tile` = x + y *
 * grid[0] + z * grid[0] * grid[1]. Every ROOT transform goes to the cell its position falls in, every descendant follows
 * its root (no parent chain is cut: a tile computes the world's matrices bit for bit), a mesh follows its entity's
 * transform; free slots and meshes without a transform go to tile 0. Inside the tile slots keep their order, entity ids
 * are renumbered from 1, parents remapped. The same rule as garden_amd/multi.py::partition_world (tests compare them).
 * Each rank of a multi-GPU run parses the scene, keeps its own tile, binds it and culls; gv_scene_tile_maps gives the
 * tile-local -> world slot tables (what a gathered visible index means). Destroy with gv_scene_destroy. */
int gv_scene_extract_tile(const GvScene* scene, const uint32_t grid[3], double side, uint32_t tile, GvScene** out_tile);
/* What ONE RANK of `world_size` owns, as one scene: the cells of the same grid are put in Morton (Z-curve) order of their
 * (x, y, z) coordinates and dealt to the ranks in rounds — every `world_size` consecutive cells of that order (a compact block
 * of space) give one cell to every rank, in an order that rotates from round to round: the k-th cell goes to rank
 * (k + ((k / world_size * 2654435761 mod 2^32) >> 16)) % world_size — and the cells of `rank` are cut out together: one pool,
 * one cull launch per rank. With many more cells than ranks (16 x 16 x 16 for 8 GPUs) every rank holds an even share of
 * whatever region a view looks at: one cell per rank leaves the ranks behind the camera idle and the frame waiting for the one
 * in front of it. The reference's split is even by construction (equal contiguous index ranges,
 * source/thread-pool.cpp:180-194). Everything else as gv_scene_extract_tile (roots decide, descendants follow, ids
 * renumbered; gv_scene_tile_maps gives the local -> world slot tables), except that free slots and meshes without a
 * transform are dealt out too (slot % world_size) instead of piling up on rank 0. grid: at most 32768 cells. */
int gv_scene_extract_rank(const GvScene* scene, const uint32_t grid[3], double side, uint32_t rank, uint32_t world_size, GvScene** out_tile);
/* The same dealing rule for positions an engine holds itself: owners[i] = the rank whose cells contain position i (3 floats at
 * positions + i * stride bytes). Ownership is a matter of BALANCE, not of correctness — an entity is culled the same wherever it
 * lives — so a moving scene re-bins at its leisure: compare a root's owner with the rank it lives on now and then, and migrate
 * (destroy there, create here: the pools' own dirty marks) only the roots that have crossed into another rank's cell (SURVEY.md
 * §8e: "re-bin only roots whose position crosses a cell"). Host-only, no context needed. */
int gv_cell_owner(const uint32_t grid[3], double side, uint32_t world_size, const float* positions, uint32_t stride, uint32_t count,
                  uint32_t* owners);
int gv_scene_tile_maps(const GvScene* tile, uint32_t pool_id, const uint32_t** transform_global, uint32_t* transform_count,
                       const uint32_t** mesh_global, uint32_t* mesh_count);

/* ---- world matrices: TransformComponent::calcModel() with cameraPosition = 0 for every transform
 * slot (transform.hpp:197-214), cached on the device ---- */
typedef enum GvSweepMode {
    GV_SWEEP_VALU = 0, /* v_fma_f32 chain */
    GV_SWEEP_MFMA = 1, /* v_mfma_f32_4x4x1_16b_f32 chain (bit-identical; self-tested at gv_create) */
    /* Deferred: the NEXT gv_cull also produces the world matrices. When its pool is exactly paired with the transform
     * pool (mesh entry i belongs to transform slot i) the MFMA sweep and the cull of view 0 run as one pass over the
     * streams; otherwise the MFMA sweep is launched in front of the ordinary cull. Same bits either way. */
    GV_SWEEP_WITH_CULL = 2,
    GV_SWEEP_WITH_CULL_VALU = 3, /* the same with the v_fma_f32 chain */
    /* Brings the cache up to date with the least work (same bits as a full sweep): nothing is launched when no transform
     * changed since the last sweep; after gv_mark_dirty ranges (GV_DIRTY_TRANSFORM, ranged GV_DIRTY_HIERARCHY) only the
     * slots whose parent chain contains a re-mirrored transform are recomputed — a dirty subtree, not the pool
     * (setPosition / setParent of the reference invalidate nothing: calcModel is lazy, transform.hpp:197-214; this is the
     * device-side counterpart for a cache that is kept); a full sweep otherwise (first use, entities created or
     * destroyed, most of the pool dirty). */
    GV_SWEEP_INCREMENTAL = 4
} GvSweepMode;
int gv_sweep(GvCtx* ctx, uint32_t mode);
/* Reads back `count` world matrices (12 floats each, float4x3 order) starting at transform slot `first`. */
int gv_get_world(GvCtx* ctx, uint32_t first, uint32_t count, float* out12);

/* ---- Hi-Z: replaces HizRenderSystem::downsampleHiz (source/system/render/hiz.cpp:104-167,
 * shaders/hiz.frag:23-63). depth = reversed-Z fp32, row-major, `width` x `height`. ---- */
typedef enum GvMemKind { GV_MEM_HOST = 0, GV_MEM_DEVICE = 1 } GvMemKind;
int gv_hiz_build(GvCtx* ctx, const float* depth, uint32_t width, uint32_t height, uint32_t mem_kind);
/* Re-runs the reduction on the depth image already resident from the last gv_hiz_build. */
int gv_hiz_rebuild(GvCtx* ctx);
/* Copies mip `level` (>= 1) back as (min,max) float pairs; *w, *h receive its size. */
int gv_hiz_read_level(GvCtx* ctx, uint32_t level, float* out_pairs, uint32_t* w, uint32_t* h);
int gv_hiz_mip_count(GvCtx* ctx, uint32_t* mip_count);

/* ---- lifecycle, errors, metrics ---- */
int gv_create(const GvConfig* config, GvCtx** out_ctx);
void gv_destroy(GvCtx* ctx);
/* Text of the last failure on this context (or of the last failed gv_create when ctx == NULL). */
const char* gv_last_error(const GvCtx* ctx);
uint32_t gv_abi_version(void);

typedef enum GvKernelId {
    GV_K_CULL = 0,     /* frustum (+Hi-Z) test, isVisible, tile-compacted records, per-chunk counts */
    GV_K_SCAN = 1,     /* chunk-count scan */
    GV_K_EMIT = 2,     /* order-stable compaction of the record segments */
    GV_K_HIZ = 3,      /* pyramid reduction (all launches of one build) */
    GV_K_SWEEP = 4,    /* world-matrix sweep */
    GV_K_SORT = 5,     /* radix sort of the records (all launches of one gv_sort) */
    GV_K_COUNT = 6
} GvKernelId;
typedef struct GvStats {
    uint64_t launches[GV_K_COUNT];   /* kernel launches since gv_stats_reset */
    double device_ms[GV_K_COUNT];    /* summed hipEvent durations (GV_CONFIG_PROFILE_EVENTS only) */
    uint64_t upload_bytes;           /* H2D mirror bytes since reset */
    uint32_t max_depth;              /* longest parent chain in the mirror */
    uint32_t transform_count, mesh_count[GV_MAX_POOLS];
    uint64_t bounds_blocks_total;    /* GV_CONFIG_BLOCK_BOUNDS: workgroups of the last cull that ran with boxes (0: none
                                        since gv_stats_reset) ... */
    uint64_t bounds_blocks_examined; /* ... and how many of them had to run the per-entity path */
    uint64_t mirror_reorders;        /* spatial re-orders of the transform mirror done ON the device since gv_create (an unsorted
                                        tail of created entities past 1/8 of the pool; no PCIe re-upload) */
    uint64_t record_targets_lost;    /* gv_pool_set_record_target calls since gv_create that found the PREVIOUS target's range unmapped
                                        when they let it go (the caller freed it too early; text in gv_last_error). The call itself
                                        succeeds: the new target is in place */
    uint64_t exchanges;              /* exchange frames sent since gv_stats_reset (gv_exchange_visible* / gv_exchange_views*: one per call) */
    uint64_t exchange_tail_rounds;   /* ... and how many of them needed the second, exactly sized exchange (a short prediction) */
} GvStats;
int gv_stats(GvCtx* ctx, GvStats* out);
int gv_stats_reset(GvCtx* ctx);
/* Measurement aid: with GV_CONFIG_PROFILE_EVENTS, bracket only every `every`-th launch of each kernel kind (1 = all, the
 * default; the first launch after gv_stats_reset is always one of them). A bracket is two event packets on the stream, ≈ 5 us
 * of stream time per launch on MI355X (measured: frame 0.164 ms with, 0.159 ms without); bench.py samples so that the timed
 * region is the frame rather than its instrumentation. GvStats.device_ms then sums the bracketed launches only;
 * gv_profile_samples tells how many there were per kind since gv_stats_reset (divide by those, not by launches). */
int gv_profile_sampling(GvCtx* ctx, uint32_t every);
/* Which kernel kinds are bracketed from now on: bit k = GvKernelId k (0: none). gv_create derives the initial mask from the
 * config flags (PROFILE_EVENTS: all, + PROFILE_CULL_ONLY: GV_K_CULL); bench.py keeps the timed region on the dominant
 * kernel and takes the per-kernel breakdown of a frame from a few extra frames outside it. */
int gv_profile_kernels(GvCtx* ctx, uint32_t kernel_mask);
int gv_profile_samples(GvCtx* ctx, uint64_t samples[GV_K_COUNT]);

/* Measurement aid (bench.py `roofline.measured_stream_peak`): `launches` read-only passes over the five input streams
 * the cull kernel reads from pool `pool_id` (mesh a/b, transform ab/c/flags = 65 bytes per entry), with the cull's own
 * loads and launch geometry, each timed with hipEvents on the context's stream. *gb_per_s = 65 * entries / median
 * launch time: the read rate this box's HBM delivers to that access pattern. Synchronises. */
int gv_debug_stream_peak(GvCtx* ctx, uint32_t pool_id, uint32_t launches, double* gb_per_s);
/* The HIP stream all work is enqueued on (hipStream_t as void*), for callers timing with their own events. */
void* gv_stream(GvCtx* ctx);
/* The library's parked host workers for a caller's own O(N) passes over its pools (the reference runs such loops on its
 * ThreadPool, source/thread-pool.cpp:173-200: contiguous ranges, the calling thread takes part): fn(user, lo, hi) over contiguous
 * pieces of [first, first + count); one piece on the calling thread for short ranges. Returns when all pieces are done. */
void gv_host_parallel_ranges(uint32_t first, uint32_t count, void (*fn)(void* user, uint32_t lo, uint32_t hi), void* user);
/* The same workers for `count` independent tasks of some size each (fn(user, task), task = 0 .. count - 1; the calling thread takes
 * part; one task: on the calling thread). A task must not call back into gv_host_parallel_* (one run at a time per process). */
void gv_host_parallel_tasks(uint32_t count, void (*fn)(void* user, uint32_t task), void* user);

#ifdef __cplusplus
}
#endif
#endif
