/*
 * gv_oracle.h — CPU restatement (ORACLE) of Garden's per-frame visibility hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, link, import or call anything in oracle/. The product
 * (garden_amd/, include/garden_vis.h, libgarden_vis.so) never includes or links this.
 *
 * PARITY UNPINNED. The arithmetic of this path (calcModel, f32x4x4 operator*, translate, Frustum,
 * isBehindFrustum, Aabb) lives in the un-vendored git submodules cfnptr/math and cfnptr/ecsm
 * (/root/reference/.gitmodules:1-3,25-27): both directories are empty in the checkout and carry no
 * pinned SHA, and the reference has no tests, golden vectors or fixtures.
 * The reference translation units on this path (source/system/render/mesh.cpp, transform.hpp) do
 * not compile here (generated garden/defines.hpp, ecsm.hpp, the math headers, Vulkan headers missing), so
 * there is no oracle/_ref build either. What IS restated from files that are present:
 *   - control flow, exits, what is written on each exit, output record:
 *       source/system/render/mesh.cpp:111-184 (unsorted) and :187-262 (sorted twin)
 *   - parent-chain association order and camera-relative translate:
 *       include/garden/system/transform.hpp:197-214, isActive :110
 *   - struct layouts: include/garden/system/render/mesh.hpp:45-55,191-205;
 *       include/garden/system/transform.hpp:31-61
 *   - thread range split: source/thread-pool.cpp:173-200
 *   - Hi-Z pyramid reduction rule: shaders/hiz.frag:23-63, sizes source/system/render/hiz.cpp:24-57
 * What is DEFINED here because upstream is absent (documented op order, explicit fmaf):
 *   calcModel, 4x4 product, plane extraction, 8-corner plane test, Hi-Z occlusion query (the
 *   reference has no occlusion query at all — SURVEY.md F3, spec §8a-7').
 *
 * All matrices are column-major float[16] (c0..c3), column-vector convention M*v, quat = xyzw
 * (include/garden/system/physics-impl.hpp:45-63).
 */
#ifndef GV_ORACLE_H
#define GV_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVO_NONE 0xFFFFFFFFu

/* Byte layout of the reference's MeshRenderComponent pool (mesh.hpp:45-55). */
typedef struct GvoMeshPool {
    uint8_t* base;           /* LinearPool::getData() */
    size_t stride;           /* IMeshRenderSystem::getMeshComponentSize() */
    uint32_t occupancy;      /* LinearPool::getOccupancy() */
    uint32_t off_entity;     /* Component::entity (u32, 0 = free slot) */
    uint32_t off_is_enabled; /* volatile bool isEnabled */
    uint32_t off_is_visible; /* volatile bool isVisible (written on main pass) */
    uint32_t off_aabb_min;   /* f32x4 */
    uint32_t off_aabb_max;   /* f32x4 */
    /* optional: what a derived system's getReadyMeshesAsync returns once the frustum test has passed (sprite.cpp:90-97,
     * ui/label.cpp:271-276): element i at ready_base + i * ready_stride, ready_width 1 or 4 bytes; NULL = 1 for all */
    const uint8_t* ready_base;
    size_t ready_stride;
    uint32_t ready_width;
} GvoMeshPool;

/* Byte layout of the reference's TransformComponent pool (transform.hpp:31-61). */
typedef struct GvoTransformPool {
    const uint8_t* base;
    size_t stride;
    uint32_t occupancy;
    uint32_t off_entity;
    uint32_t off_parent;   /* ID<Entity> parent (u32, 0 = none) */
    uint32_t off_position; /* f32x4 posChildCount (w = child count bits) */
    uint32_t off_scale;    /* f32x4 scaleChildCap (w = child capacity bits) */
    uint32_t off_rotation; /* quat xyzw */
    uint32_t off_self_active;
    uint32_t off_ancestors_active;
    uint32_t off_model_with_ancestors;
    /* Manager::tryGet<TransformComponent>(entity): entity id -> transform slot or GVO_NONE */
    const uint32_t* entity_to_transform;
    uint32_t entity_capacity;
} GvoTransformPool;

typedef struct GvoFrustum {
    float planes[6][4]; /* normalised (a,b,c,d); first `count` valid */
    uint32_t count;
} GvoFrustum;

typedef struct GvoHiz {
    const float* depth;    /* mip 0 = depth image itself (min == max == d) */
    const float* mips;     /* levels 1..mip_count-1, (min,max) float pairs, level k at mip_offset[k] pairs */
    uint32_t width, height, mip_count;
    uint32_t mip_w[16], mip_h[16];
    uint64_t mip_offset[16]; /* in (min,max) pairs from `mips`; [0] unused */
} GvoHiz;

typedef struct GvoView {
    float view_proj[16];
    float camera_position[4];
    float camera_offset[4];
    int8_t shadow_pass;   /* < 0 = main pass (writes isVisible), mesh.cpp:121 */
    uint8_t use_hiz;      /* run the build-defined occlusion query after the frustum test */
    uint8_t distance_2d;  /* sorted twin key: translation.z + 1 (mesh.cpp:250) */
    uint8_t reserved;
} GvoView;

/* Output record arrays (SoA form of UnsortedMesh/SortedMesh, mesh.hpp:191-205). */
typedef struct GvoCullOut {
    uint32_t* visible_idx;  /* pool slot i; componentOffset = i * stride */
    float* baked_model;     /* 12 floats per record: c0.xyz c1.xyz c2.xyz c3.xyz ((float4x3)model) */
    float* distance_sq;
    uint32_t draw_count;
    uint32_t instance_count;
} GvoCullOut;

enum { GVO_HIZ_RULE_REFERENCE = 0, GVO_HIZ_RULE_CONSERVATIVE = 1,
       /* OR-ed into `rule`: levels >= 1 hold what an RG16F image (hiz.hpp:41) can hold, min rounded toward -inf and max
        * toward +inf (level 0 is the depth image itself); the arrays stay float, every value is half-representable */
       GVO_HIZ_FORMAT_RG16F = 0x100 };

/* ---- math (build-defined canonical op order) ---- */
void gvo_calc_model(const float pos[3], const float rot[4], const float scale[3], float out[16]);
void gvo_mul4x4(const float a[16], const float b[16], float out[16]);
void gvo_frustum_from_view_proj(const float vp[16], GvoFrustum* out);
int gvo_is_behind_frustum(const GvoFrustum* f, const float aabb_min[3], const float aabb_max[3], const float model[16]);

/* TransformComponent::calcModel(cameraPosition), transform.hpp:197-214. */
void gvo_transform_calc_model(const GvoTransformPool* tp, uint32_t slot, const float camera_position[3], float out[16]);
/* World matrices of slots [first, first+count) with cameraPosition = 0; out = 12 floats each. */
void gvo_world_matrices(const GvoTransformPool* tp, uint32_t first, uint32_t count, float* out12);
void gvo_world_matrices_mt(const GvoTransformPool* tp, uint32_t first, uint32_t count, float* out12, uint32_t threads);

/* ---- Hi-Z ---- */
/* float -> binary16 bits rounded toward +inf (up != 0) or -inf, and the exact way back */
uint16_t gvo_half_directed(float f, int up);
float gvo_half_to_float(uint16_t half_bits);
uint32_t gvo_calc_mip_count(uint32_t w, uint32_t h);
/* Fills mip_w/h/offset; returns total (min,max) pairs needed for levels >= 1. */
uint64_t gvo_hiz_layout(uint32_t w, uint32_t h, GvoHiz* out);
/* hiz.frag:23-63: builds levels 1.. into `mips` (hiz->mips must point to writable storage). */
void gvo_hiz_build(GvoHiz* hiz, float* mips, int rule);
/* Same pyramid, the rows of each level split over `threads` threads like ThreadPool::addItems (bench.py cpu_baseline). */
void gvo_hiz_build_mt(GvoHiz* hiz, float* mips, int rule, uint32_t threads);
/* Build-defined occlusion query (SURVEY.md §8a-7'); returns 1 if occluded. */
int gvo_hiz_occluded(const GvoHiz* hiz, const float view_proj[16], const float aabb_min[3],
                     const float aabb_max[3], const float model[16]);

/* ---- the hot loop: prepareUnsortedMeshes / prepareSortedMeshes, mesh.cpp:111-262 ---- */
/* Items [item_offset, item_end) — note item_end is an END index (thread-pool.cpp:186-187). */
void gvo_prepare_meshes_range(const GvoMeshPool* mp, const GvoTransformPool* tp, const GvoView* view,
                              const GvoFrustum* frustum, const GvoHiz* hiz, uint32_t item_offset,
                              uint32_t item_end, GvoCullOut* out);
/* prepareMeshes dispatch over ThreadPool::addItems ranges (mesh.cpp:498-503, thread-pool.cpp:173-200).
 * threads == 1 runs inline. out arrays must hold `occupancy` records. Order across ranges follows
 * fetch_add arrival, as in the reference (nondeterministic for threads > 1). */
void gvo_prepare_meshes(const GvoMeshPool* mp, const GvoTransformPool* tp, const GvoView* view,
                        const GvoHiz* hiz, uint32_t threads, GvoCullOut* out);

/* ---- AVX2 form (gv_oracle_avx2.c): 8 entities per iteration over an SoA copy of the pools; bit-identical to the
 * scalar routines above; the timed cpu_baseline of bench.py ---- */
typedef struct GvoSoa GvoSoa;
GvoSoa* gvo_soa_build(const GvoMeshPool* mp, const GvoTransformPool* tp);
/* the same, filled by `threads` workers over the ranges gvo_prepare_meshes_avx2 will hand them (first touch = the culling thread) */
GvoSoa* gvo_soa_build_threads(const GvoMeshPool* mp, const GvoTransformPool* tp, uint32_t threads);
void gvo_soa_free(GvoSoa* soa);
void gvo_prepare_meshes_avx2(const GvoSoa* soa, const GvoMeshPool* mp, const GvoView* view, const GvoHiz* hiz,
                             uint32_t threads, GvoCullOut* out);

/* sortMeshes (mesh.cpp:265-328): ascending distanceSq (unsorted buffers) or descending (sorted). */
void gvo_sort_records(GvoCullOut* out, int descending);

#ifdef __cplusplus
}
#endif
#endif
