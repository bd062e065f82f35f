/*
 * gv_oracle_avx2.c — AVX2+FMA (8 entities per iteration) form of the CPU restatement, over an SoA copy of the
 * pools. TEST INFRASTRUCTURE ONLY (same rules as gv_oracle.c): it is the timed `cpu_baseline` of bench.py — the
 * "reference AVX2 CPU path" the reference itself cannot provide here (its sources do not build) —
 * and it must agree with the scalar oracle bit for bit (tests/test_oracle_avx2.py).
 *
 * Same control flow as prepareUnsortedMeshes (source/system/render/mesh.cpp:137-175), same canonical arithmetic
 * (explicit fused multiply-adds in the written order), same ThreadPool::addItems range split
 * (source/thread-pool.cpp:173-200). The Hi-Z query of the frustum survivors runs 8-wide too (hiz_occluded8: the scalar routine's
 * operations lane by lane, texels by gather).
 * Build flags: -O2 -march=haswell -ffp-contract=off (cmake/compile-options.cmake:34-36 uses -march=haswell too).
 */
#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#include "gv_oracle.h"

void gvo_pool_run(void* (*fn)(void*), void** args, int count, int threads); /* gv_oracle.c */

typedef struct GvoSoa {
    uint32_t mesh_count, xf_count;
    /* transforms, indexed by transform slot */
    float *px, *py, *pz, *qx, *qy, *qz, *qw, *sx, *sy, *sz;
    int32_t* parent;  /* parent slot or -1 */
    uint8_t* xf_flags; /* bit0 active, bit1 modelWithAncestors */
    /* meshes, indexed by mesh slot */
    float *mnx, *mny, *mnz, *mxx, *mxy, *mxz;
    int32_t* slot;      /* transform slot or -1 */
    uint8_t* candidate; /* entity != 0 && isEnabled */
} GvoSoa;

static void* xmalloc(size_t n) { void* p = NULL; if (posix_memalign(&p, 64, n ? n : 64)) abort(); return p; }

/* One contiguous range of transform slots / mesh slots copied into the SoA arrays. Run by the thread that will later cull
 * the same range (gvo_soa_build_threads), so that the pages it touches first live on its own NUMA node. */
typedef struct SoaFill {
    GvoSoa* s; const GvoMeshPool* mp; const GvoTransformPool* tp;
    uint32_t xlo, xhi, mlo, mhi;
} SoaFill;

static void* soa_fill_main(void* arg)
{
    const SoaFill* f = (const SoaFill*)arg;
    GvoSoa* s = f->s; const GvoMeshPool* mp = f->mp; const GvoTransformPool* tp = f->tp;
    const uint32_t nx = tp->occupancy;
    for (uint32_t t = f->xlo; t < f->xhi; t++) {
        const uint8_t* c = tp->base + (size_t)t * tp->stride;
        const float* p = (const float*)(c + tp->off_position);
        const float* q = (const float*)(c + tp->off_rotation);
        const float* sc = (const float*)(c + tp->off_scale);
        s->px[t] = p[0]; s->py[t] = p[1]; s->pz[t] = p[2];
        s->qx[t] = q[0]; s->qy[t] = q[1]; s->qz[t] = q[2]; s->qw[t] = q[3];
        s->sx[t] = sc[0]; s->sy[t] = sc[1]; s->sz[t] = sc[2];
        uint32_t pe; memcpy(&pe, c + tp->off_parent, 4);
        uint32_t ps = (pe == 0 || pe >= tp->entity_capacity) ? GVO_NONE : tp->entity_to_transform[pe];
        s->parent[t] = (ps == GVO_NONE || ps >= nx) ? -1 : (int32_t)ps;
        s->xf_flags[t] = (uint8_t)(((c[tp->off_self_active] && c[tp->off_ancestors_active]) ? 1 : 0) |
                                   (c[tp->off_model_with_ancestors] ? 2 : 0));
    }
    for (uint32_t i = f->mlo; i < f->mhi; i++) {
        const uint8_t* c = mp->base + (size_t)i * mp->stride;
        const float* a = (const float*)(c + mp->off_aabb_min);
        const float* b = (const float*)(c + mp->off_aabb_max);
        s->mnx[i] = a[0]; s->mny[i] = a[1]; s->mnz[i] = a[2];
        s->mxx[i] = b[0]; s->mxy[i] = b[1]; s->mxz[i] = b[2];
        uint32_t e; memcpy(&e, c + mp->off_entity, 4);
        uint32_t ts = (e == 0 || e >= tp->entity_capacity) ? GVO_NONE : tp->entity_to_transform[e];
        s->slot[i] = (ts == GVO_NONE || ts >= nx) ? -1 : (int32_t)ts;
        s->candidate[i] = (uint8_t)((e != 0 && c[mp->off_is_enabled]) ? 1 : 0);
    }
    return NULL;
}

/* threads > 1: the arrays are filled (first touched) by `threads` workers over the SAME contiguous ranges
 * gvo_prepare_meshes_avx2 hands them (ThreadPool::addItems' split, thread-pool.cpp:179-194). */
GvoSoa* gvo_soa_build_threads(const GvoMeshPool* mp, const GvoTransformPool* tp, uint32_t threads)
{
    GvoSoa* s = (GvoSoa*)calloc(1, sizeof(GvoSoa));
    const uint32_t nm = mp->occupancy, nx = tp->occupancy;
    const size_t pm = ((size_t)nm + 8) * 4, px = ((size_t)nx + 8) * 4;
    s->mesh_count = nm; s->xf_count = nx;
    float** xf_f[] = {&s->px, &s->py, &s->pz, &s->qx, &s->qy, &s->qz, &s->qw, &s->sx, &s->sy, &s->sz};
    for (int k = 0; k < 10; k++) { *xf_f[k] = (float*)xmalloc(px); memset(*xf_f[k] + nx, 0, 32); }  /* the 8-lane tail only */
    s->parent = (int32_t*)xmalloc(px); s->xf_flags = (uint8_t*)xmalloc(nx + 8);
    float** m_f[] = {&s->mnx, &s->mny, &s->mnz, &s->mxx, &s->mxy, &s->mxz};
    for (int k = 0; k < 6; k++) { *m_f[k] = (float*)xmalloc(pm); memset(*m_f[k] + nm, 0, 32); }
    s->slot = (int32_t*)xmalloc(pm); s->candidate = (uint8_t*)xmalloc(nm + 8);
    memset(s->parent + nx, 0xFF, 32); memset(s->xf_flags + nx, 0, 8);
    memset(s->slot + nm, 0xFF, 32); memset(s->candidate + nm, 0, 8);
    const uint32_t most = nm > nx ? nm : nx;
    uint32_t tasks = threads < 1 ? 1 : threads;
    if (tasks > most) tasks = most ? most : 1;
    SoaFill* fills = (SoaFill*)calloc(tasks, sizeof(SoaFill));
    void** argv = (void**)malloc(sizeof(void*) * tasks);
    const uint32_t mper = (uint32_t)ceilf((float)nm / (float)tasks), xper = (uint32_t)ceilf((float)nx / (float)tasks);
    for (uint32_t k = 0; k < tasks; k++) {
        SoaFill* f = &fills[k];
        f->s = s; f->mp = mp; f->tp = tp;
        f->mlo = mper * k < nm ? mper * k : nm; f->mhi = f->mlo + mper < nm ? f->mlo + mper : nm;
        f->xlo = xper * k < nx ? xper * k : nx; f->xhi = f->xlo + xper < nx ? f->xlo + xper : nx;
        argv[k] = f;
    }
    if (tasks == 1)
        soa_fill_main(argv[0]);
    else
        gvo_pool_run(soa_fill_main, argv, (int)tasks, (int)tasks);
    free(argv); free(fills);
    return s;
}

GvoSoa* gvo_soa_build(const GvoMeshPool* mp, const GvoTransformPool* tp) { return gvo_soa_build_threads(mp, tp, 1); }

void gvo_soa_free(GvoSoa* s)
{
    if (!s) return;
    float* f[] = {s->px, s->py, s->pz, s->qx, s->qy, s->qz, s->qw, s->sx, s->sy, s->sz, s->mnx, s->mny, s->mnz, s->mxx, s->mxy, s->mxz};
    for (unsigned k = 0; k < sizeof(f) / sizeof(f[0]); k++) free(f[k]);
    free(s->parent); free(s->xf_flags); free(s->slot); free(s->candidate); free(s);
}

typedef struct M34 { __m256 c0x, c0y, c0z, c1x, c1y, c1z, c2x, c2y, c2z, c3x, c3y, c3z; } M34;

#define FMA(a, b, c) _mm256_fmadd_ps((a), (b), (c))
#define MUL(a, b) _mm256_mul_ps((a), (b))
#define ADD(a, b) _mm256_add_ps((a), (b))
#define SUB(a, b) _mm256_sub_ps((a), (b))

/* calcModel: the same written order as gvo_calc_model */
static inline M34 calc_model8(__m256 px, __m256 py, __m256 pz, __m256 x, __m256 y, __m256 z, __m256 w, __m256 sx, __m256 sy, __m256 sz)
{
    const __m256 one = _mm256_set1_ps(1.0f), sign = _mm256_set1_ps(-0.0f);
    const __m256 x2 = ADD(x, x), y2 = ADD(y, y), z2 = ADD(z, z);
    const __m256 zz = MUL(z, z2), yy = MUL(y, y2);
    const __m256 wx = MUL(w, x2), wy = MUL(w, y2), wz = MUL(w, z2);
    const __m256 nwx = _mm256_xor_ps(wx, sign), nwy = _mm256_xor_ps(wy, sign), nwz = _mm256_xor_ps(wz, sign);
    const __m256 r00 = SUB(one, FMA(y, y2, zz)), r11 = SUB(one, FMA(x, x2, zz)), r22 = SUB(one, FMA(x, x2, yy));
    const __m256 r10 = FMA(x, y2, wz), r01 = FMA(x, y2, nwz);
    const __m256 r20 = FMA(x, z2, nwy), r02 = FMA(x, z2, wy);
    const __m256 r21 = FMA(y, z2, wx), r12 = FMA(y, z2, nwx);
    M34 m;
    m.c0x = MUL(r00, sx); m.c0y = MUL(r10, sx); m.c0z = MUL(r20, sx);
    m.c1x = MUL(r01, sy); m.c1y = MUL(r11, sy); m.c1z = MUL(r21, sy);
    m.c2x = MUL(r02, sz); m.c2y = MUL(r12, sz); m.c2z = MUL(r22, sz);
    m.c3x = px; m.c3y = py; m.c3z = pz;
    return m;
}

/* one element of a*b: fma chain k = 0..3 from +0; b3 = 0 or 1 */
static inline __m256 mul_elem8(__m256 a0, __m256 a1, __m256 a2, __m256 a3, __m256 b0, __m256 b1, __m256 b2, __m256 b3)
{
    __m256 acc = FMA(a0, b0, _mm256_setzero_ps());
    acc = FMA(a1, b1, acc);
    acc = FMA(a2, b2, acc);
    return FMA(a3, b3, acc);
}

static inline M34 mul_affine8(const M34* a, const M34* b)
{
    const __m256 zero = _mm256_setzero_ps(), one = _mm256_set1_ps(1.0f);
    M34 r;
    r.c0x = mul_elem8(a->c0x, a->c1x, a->c2x, a->c3x, b->c0x, b->c0y, b->c0z, zero);
    r.c0y = mul_elem8(a->c0y, a->c1y, a->c2y, a->c3y, b->c0x, b->c0y, b->c0z, zero);
    r.c0z = mul_elem8(a->c0z, a->c1z, a->c2z, a->c3z, b->c0x, b->c0y, b->c0z, zero);
    r.c1x = mul_elem8(a->c0x, a->c1x, a->c2x, a->c3x, b->c1x, b->c1y, b->c1z, zero);
    r.c1y = mul_elem8(a->c0y, a->c1y, a->c2y, a->c3y, b->c1x, b->c1y, b->c1z, zero);
    r.c1z = mul_elem8(a->c0z, a->c1z, a->c2z, a->c3z, b->c1x, b->c1y, b->c1z, zero);
    r.c2x = mul_elem8(a->c0x, a->c1x, a->c2x, a->c3x, b->c2x, b->c2y, b->c2z, zero);
    r.c2y = mul_elem8(a->c0y, a->c1y, a->c2y, a->c3y, b->c2x, b->c2y, b->c2z, zero);
    r.c2z = mul_elem8(a->c0z, a->c1z, a->c2z, a->c3z, b->c2x, b->c2y, b->c2z, zero);
    r.c3x = mul_elem8(a->c0x, a->c1x, a->c2x, a->c3x, b->c3x, b->c3y, b->c3z, one);
    r.c3y = mul_elem8(a->c0y, a->c1y, a->c2y, a->c3y, b->c3x, b->c3y, b->c3z, one);
    r.c3z = mul_elem8(a->c0z, a->c1z, a->c2z, a->c3z, b->c3x, b->c3y, b->c3z, one);
    return r;
}

static inline M34 blend_m34(const M34* old, const M34* neu, __m256 mask)
{
    M34 r;
    const __m256* o = &old->c0x; const __m256* n = &neu->c0x; __m256* d = &r.c0x;
    for (int k = 0; k < 12; k++) d[k] = _mm256_blendv_ps(o[k], n[k], mask);
    return r;
}

static inline M34 gather_model8(const GvoSoa* s, __m256i idx, __m256 mask)
{
    const __m256 z = _mm256_setzero_ps();
#define G(arr) _mm256_mask_i32gather_ps(z, (arr), idx, mask, 4)
    return calc_model8(G(s->px), G(s->py), G(s->pz), G(s->qx), G(s->qy), G(s->qz), G(s->qw), G(s->sx), G(s->sy), G(s->sz));
#undef G
}

/* gvo_hiz_occluded for 8 boxes at once (their 8 corners each are already in registers): the same operations in the same
 * order per lane — IEEE divisions, fminf / fmaxf as glibc defines them (a NaN operand is ignored), the integer level search,
 * the four texel reads as gathers — so that the result equals the scalar routine's bit for bit (tests/test_oracle_avx2.py).
 * `lanes`: bit l = box l is to be tested; returns the mask of occluded boxes. Needs every texel index to fit 31 bits. */
static inline __m256 fmin8(__m256 acc, __m256 u)
{   /* fminf(acc, u): acc < u ? acc : (u is NaN ? acc : u); MINPS returns its second operand when either is NaN */
    return _mm256_blendv_ps(_mm256_min_ps(acc, u), acc, _mm256_cmp_ps(u, u, _CMP_UNORD_Q));
}
static inline __m256 fmax8(__m256 acc, __m256 u)
{
    return _mm256_blendv_ps(_mm256_max_ps(acc, u), acc, _mm256_cmp_ps(u, u, _CMP_UNORD_Q));
}
static inline __m256 clamp01_8(__m256 a)
{   /* a > 0 ? (a < 1 ? a : 1) : 0 */
    const __m256 one = _mm256_set1_ps(1.0f);
    const __m256 t = _mm256_blendv_ps(one, a, _mm256_cmp_ps(a, one, _CMP_LT_OQ));
    return _mm256_and_ps(t, _mm256_cmp_ps(a, _mm256_setzero_ps(), _CMP_GT_OQ));
}
static int hiz_fits_int32(const GvoHiz* hz)
{
    const uint64_t last = hz->mip_count ? hz->mip_offset[hz->mip_count - 1] + (uint64_t)hz->mip_w[hz->mip_count - 1] * hz->mip_h[hz->mip_count - 1] : 0;
    return (uint64_t)hz->width * hz->height < (1ull << 31) && 2 * last < (1ull << 31);
}
static inline uint32_t hiz_occluded8(const GvoHiz* hz, const float vp[16], const __m256* cx, const __m256* cy, const __m256* cz, uint32_t lanes)
{
    const __m256 zero = _mm256_setzero_ps(), half = _mm256_set1_ps(0.5f), one = _mm256_set1_ps(1.0f);
    __m256 umin = zero, umax = zero, vmin = zero, vmax = zero, znear = zero;
    __m256 bounded = _mm256_castsi256_ps(_mm256_set1_epi32(-1));
    for (int k = 0; k < 8; k++) {
        const __m256 clx = FMA(_mm256_set1_ps(vp[0]), cx[k], FMA(_mm256_set1_ps(vp[4]), cy[k], FMA(_mm256_set1_ps(vp[8]), cz[k], _mm256_set1_ps(vp[12]))));
        const __m256 cly = FMA(_mm256_set1_ps(vp[1]), cx[k], FMA(_mm256_set1_ps(vp[5]), cy[k], FMA(_mm256_set1_ps(vp[9]), cz[k], _mm256_set1_ps(vp[13]))));
        const __m256 clz = FMA(_mm256_set1_ps(vp[2]), cx[k], FMA(_mm256_set1_ps(vp[6]), cy[k], FMA(_mm256_set1_ps(vp[10]), cz[k], _mm256_set1_ps(vp[14]))));
        const __m256 clw = FMA(_mm256_set1_ps(vp[3]), cx[k], FMA(_mm256_set1_ps(vp[7]), cy[k], FMA(_mm256_set1_ps(vp[11]), cz[k], _mm256_set1_ps(vp[15]))));
        bounded = _mm256_and_ps(bounded, _mm256_cmp_ps(clw, zero, _CMP_GT_OQ)); /* !(clw > 0): cannot bound -> visible */
        const __m256 rcp = _mm256_div_ps(one, clw);
        const __m256 u = FMA(MUL(clx, rcp), half, half), v = FMA(MUL(cly, rcp), half, half), zc = MUL(clz, rcp);
        if (k == 0) {
            umin = umax = u; vmin = vmax = v; znear = zc;
        } else {
            umin = fmin8(umin, u); umax = fmax8(umax, u);
            vmin = fmin8(vmin, v); vmax = fmax8(vmax, v);
            znear = fmax8(znear, zc);
        }
    }
    lanes &= (uint32_t)_mm256_movemask_ps(bounded);
    if (!lanes)
        return 0;
    umin = clamp01_8(umin); umax = clamp01_8(umax); vmin = clamp01_8(vmin); vmax = clamp01_8(vmax);
    const __m256 Wf = _mm256_set1_ps((float)(int)hz->width), Hf = _mm256_set1_ps((float)(int)hz->height);
    const __m256i wm1 = _mm256_set1_epi32((int)hz->width - 1), hm1 = _mm256_set1_epi32((int)hz->height - 1);
    /* lanes that are not tested may hold anything (NaN, huge): keep their integers harmless */
    const __m256 live = _mm256_castsi256_ps(_mm256_cmpgt_epi32(_mm256_and_si256(_mm256_set1_epi32((int)lanes), _mm256_setr_epi32(1, 2, 4, 8, 16, 32, 64, 128)),
                                                              _mm256_setzero_si256()));
    const __m256i ix0 = _mm256_min_epi32(_mm256_cvttps_epi32(_mm256_and_ps(MUL(umin, Wf), live)), wm1);
    const __m256i ix1 = _mm256_min_epi32(_mm256_cvttps_epi32(_mm256_and_ps(MUL(umax, Wf), live)), wm1);
    const __m256i iy0 = _mm256_min_epi32(_mm256_cvttps_epi32(_mm256_and_ps(MUL(vmin, Hf), live)), hm1);
    const __m256i iy1 = _mm256_min_epi32(_mm256_cvttps_epi32(_mm256_and_ps(MUL(vmax, Hf), live)), hm1);
    /* smallest level at which the pixel rect touches <= 2x2 texels */
    __m256i level = _mm256_setzero_si256(), still = _mm256_castps_si256(live);
    const __m256i one_i = _mm256_set1_epi32(1);
    for (uint32_t L = 0; L + 1 < hz->mip_count; L++) {
        const __m128i sh = _mm_cvtsi32_si128((int)L);
        const __m256i dx = _mm256_sub_epi32(_mm256_srl_epi32(ix1, sh), _mm256_srl_epi32(ix0, sh));
        const __m256i dy = _mm256_sub_epi32(_mm256_srl_epi32(iy1, sh), _mm256_srl_epi32(iy0, sh));
        still = _mm256_and_si256(still, _mm256_or_si256(_mm256_cmpgt_epi32(dx, one_i), _mm256_cmpgt_epi32(dy, one_i)));
        if (!_mm256_movemask_ps(_mm256_castsi256_ps(still)))
            break;
        level = _mm256_sub_epi32(level, still); /* still = -1 where the rect is still too wide at L: level = L + 1 */
    }
    int off32[16];
    for (int k = 0; k < 16; k++) off32[k] = (int)hz->mip_offset[k];
    const __m256i lw = _mm256_i32gather_epi32((const int*)hz->mip_w, level, 4), lh = _mm256_i32gather_epi32((const int*)hz->mip_h, level, 4);
    const __m256i lwm1 = _mm256_sub_epi32(lw, one_i), lhm1 = _mm256_sub_epi32(lh, one_i);
    const __m256i tx0 = _mm256_min_epi32(_mm256_srlv_epi32(ix0, level), lwm1), tx1 = _mm256_min_epi32(_mm256_srlv_epi32(ix1, level), lwm1);
    const __m256i ty0 = _mm256_min_epi32(_mm256_srlv_epi32(iy0, level), lhm1), ty1 = _mm256_min_epi32(_mm256_srlv_epi32(iy1, level), lhm1);
    const __m256i is0 = _mm256_cmpeq_epi32(level, _mm256_setzero_si256());
    const __m256 is0f = _mm256_and_ps(_mm256_castsi256_ps(is0), live), isnf = _mm256_andnot_ps(_mm256_castsi256_ps(is0), live);
    const __m256i base = _mm256_i32gather_epi32(off32, level, 4);
#define TEXEL(TX, TY) ({                                                                                                        \
        const __m256i at = _mm256_add_epi32(_mm256_mullo_epi32((TY), lw), (TX));                                                    \
        const __m256 d = _mm256_mask_i32gather_ps(zero, hz->depth, at, is0f, 4);                                                    \
        const __m256 m = _mm256_mask_i32gather_ps(zero, hz->mips, _mm256_slli_epi32(_mm256_add_epi32(base, at), 1), isnf, 4);       \
        _mm256_blendv_ps(m, d, is0f); })
    __m256 zfar = TEXEL(tx0, ty0), a;
    a = TEXEL(tx1, ty0); zfar = _mm256_blendv_ps(zfar, a, _mm256_cmp_ps(a, zfar, _CMP_LT_OQ));
    a = TEXEL(tx0, ty1); zfar = _mm256_blendv_ps(zfar, a, _mm256_cmp_ps(a, zfar, _CMP_LT_OQ));
    a = TEXEL(tx1, ty1); zfar = _mm256_blendv_ps(zfar, a, _mm256_cmp_ps(a, zfar, _CMP_LT_OQ));
#undef TEXEL
    return lanes & (uint32_t)_mm256_movemask_ps(_mm256_cmp_ps(znear, zfar, _CMP_LT_OQ));
}

void gvo_prepare_meshes_range_avx2(const GvoSoa* s, const GvoMeshPool* mp, const GvoView* view, const GvoFrustum* fr,
                                   const GvoHiz* hiz, uint32_t item_offset, uint32_t item_end, GvoCullOut* out)
{
    const int main_pass = view->shadow_pass < 0;
    uint32_t draw_count = 0, instance_count = 0;
    const __m256 zero = _mm256_setzero_ps();
    const __m256 camx = _mm256_set1_ps(view->camera_position[0]), camy = _mm256_set1_ps(view->camera_position[1]),
                 camz = _mm256_set1_ps(view->camera_position[2]);
    const __m256i lanes = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    const int hiz_8wide = view->use_hiz && hiz && hiz_fits_int32(hiz);
    for (uint32_t i = item_offset; i < item_end; i += 8) {
        const uint32_t left = item_end - i;
        const __m256i in_range = _mm256_cmpgt_epi32(_mm256_set1_epi32((int)(left < 8 ? left : 8)), lanes);
        const __m256 mnx = _mm256_loadu_ps(s->mnx + i), mny = _mm256_loadu_ps(s->mny + i), mnz = _mm256_loadu_ps(s->mnz + i);
        const __m256 mxx = _mm256_loadu_ps(s->mxx + i), mxy = _mm256_loadu_ps(s->mxy + i), mxz = _mm256_loadu_ps(s->mxz + i);
        /* mesh.cpp:140-142 */
        const __m256 empty = _mm256_and_ps(_mm256_and_ps(_mm256_cmp_ps(SUB(mxx, mnx), zero, _CMP_LE_OQ),
                                                         _mm256_cmp_ps(SUB(mxy, mny), zero, _CMP_LE_OQ)),
                                           _mm256_cmp_ps(SUB(mxz, mnz), zero, _CMP_LE_OQ));
        const __m256i cand8 = _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i*)(s->candidate + i)));
        const __m256i slot = _mm256_loadu_si256((const __m256i*)(s->slot + i));
        __m256i ok = _mm256_and_si256(in_range, _mm256_cmpgt_epi32(cand8, _mm256_setzero_si256()));
        ok = _mm256_andnot_si256(_mm256_castps_si256(empty), ok);
        ok = _mm256_and_si256(ok, _mm256_cmpgt_epi32(slot, _mm256_set1_epi32(-1)));
        /* transform flags (isActive, modelWithAncestors) */
        int sl[8], okm[8];
        _mm256_storeu_si256((__m256i*)sl, slot); _mm256_storeu_si256((__m256i*)okm, ok);
        int active[8], with_anc[8];
        for (int l = 0; l < 8; l++) {
            const uint8_t f = okm[l] ? s->xf_flags[sl[l]] : 0;
            active[l] = (f & 1) ? -1 : 0; with_anc[l] = (f & 2) ? -1 : 0;
        }
        ok = _mm256_and_si256(ok, _mm256_loadu_si256((const __m256i*)active)); /* mesh.cpp:150 */
        int any = _mm256_movemask_ps(_mm256_castsi256_ps(ok));
        uint32_t vis_mask = 0;
        M34 m;
        memset(&m, 0, sizeof(m));
        if (any) {
            const __m256 okf = _mm256_castsi256_ps(ok);
            m = gather_model8(s, slot, okf);
            /* parent chain, transform.hpp:204-210: model = parentModel * model */
            __m256i par = _mm256_mask_i32gather_epi32(_mm256_set1_epi32(-1), s->parent, slot, ok, 4);
            par = _mm256_blendv_epi8(_mm256_set1_epi32(-1), par, _mm256_and_si256(ok, _mm256_loadu_si256((const __m256i*)with_anc)));
            for (;;) {
                const __m256i has = _mm256_cmpgt_epi32(par, _mm256_set1_epi32(-1));
                if (!_mm256_movemask_ps(_mm256_castsi256_ps(has)))
                    break;
                const __m256 hasf = _mm256_castsi256_ps(has);
                const M34 pm = gather_model8(s, par, hasf);
                const M34 prod = mul_affine8(&pm, &m);
                m = blend_m34(&m, &prod, hasf);
                par = _mm256_mask_i32gather_epi32(_mm256_set1_epi32(-1), s->parent, par, has, 4);
            }
            /* translate(-cameraPosition, model) */
            m.c3x = SUB(m.c3x, camx); m.c3y = SUB(m.c3y, camy); m.c3z = SUB(m.c3z, camz);
            /* corners + planes: behind iff some plane has all 8 distances < 0 */
            __m256 cx[8], cy[8], cz[8];
            for (int k = 0; k < 8; k++) {
                const __m256 x = (k & 1) ? mxx : mnx, y = (k & 2) ? mxy : mny, z = (k & 4) ? mxz : mnz;
                cx[k] = FMA(m.c0x, x, FMA(m.c1x, y, FMA(m.c2x, z, m.c3x)));
                cy[k] = FMA(m.c0y, x, FMA(m.c1y, y, FMA(m.c2y, z, m.c3y)));
                cz[k] = FMA(m.c0z, x, FMA(m.c1z, y, FMA(m.c2z, z, m.c3z)));
            }
            __m256 behind = zero;
            for (uint32_t p = 0; p < fr->count; p++) {
                const __m256 nx = _mm256_set1_ps(fr->planes[p][0]), ny = _mm256_set1_ps(fr->planes[p][1]);
                const __m256 nz = _mm256_set1_ps(fr->planes[p][2]), nw = _mm256_set1_ps(fr->planes[p][3]);
                __m256 all_neg = _mm256_castsi256_ps(_mm256_set1_epi32(-1));
                for (int k = 0; k < 8; k++) {
                    const __m256 d = FMA(nx, cx[k], FMA(ny, cy[k], FMA(nz, cz[k], nw)));
                    all_neg = _mm256_and_ps(all_neg, _mm256_cmp_ps(d, zero, _CMP_LT_OQ));
                }
                behind = _mm256_or_ps(behind, all_neg);
            }
            vis_mask = (uint32_t)_mm256_movemask_ps(_mm256_andnot_ps(behind, okf));
            if (vis_mask && hiz_8wide) /* the occlusion query of the frustum survivors, 8 at a time */
                vis_mask &= ~hiz_occluded8(hiz, view->view_proj, cx, cy, cz, vis_mask);
        }
        /* scalar epilogue per lane: optional Hi-Z query, isVisible write-back, record append (mesh.cpp:158-174) */
        float mm[12][8];
        if (vis_mask) {
            const __m256* src = &m.c0x;
            for (int k = 0; k < 12; k++) _mm256_storeu_ps(mm[k], src[k]);
        }
        const uint32_t lanes_here = left < 8 ? left : 8;
        for (uint32_t l = 0; l < lanes_here; l++) {
            uint8_t* mesh = mp->base + (size_t)(i + l) * mp->stride;
            int visible = (vis_mask >> l) & 1;
            if (visible && view->use_hiz && hiz && !hiz_8wide) { /* (a pyramid too large for 32-bit gather indices: the scalar routine) */
                float model[16] = {mm[0][l], mm[1][l], mm[2][l], 0, mm[3][l], mm[4][l], mm[5][l], 0,
                                   mm[6][l], mm[7][l], mm[8][l], 0, mm[9][l], mm[10][l], mm[11][l], 1};
                const float amin[3] = {s->mnx[i + l], s->mny[i + l], s->mnz[i + l]};
                const float amax[3] = {s->mxx[i + l], s->mxy[i + l], s->mxz[i + l]};
                if (gvo_hiz_occluded(hiz, view->view_proj, amin, amax, model))
                    visible = 0;
            }
            uint32_t ready_count = 1;
            if (visible && mp->ready_base) { /* derived predicate's count (sprite.cpp:90-97): 0 = not ready */
                const uint8_t* r = mp->ready_base + (size_t)(i + l) * mp->ready_stride;
                if (mp->ready_width == 4)
                    memcpy(&ready_count, r, 4);
                else
                    ready_count = *r;
                visible = ready_count != 0;
            }
            if (main_pass)
                *(mesh + mp->off_is_visible) = (uint8_t)visible;
            if (!visible)
                continue;
            out->visible_idx[draw_count] = i + l;
            float* bm = out->baked_model + (size_t)draw_count * 12;
            for (int k = 0; k < 12; k++) bm[k] = mm[k][l];
            const float tx = mm[9][l] + view->camera_offset[0], ty = mm[10][l] + view->camera_offset[1],
                        tz = mm[11][l] + view->camera_offset[2];
            out->distance_sq[draw_count] = view->distance_2d ? mm[11][l] + 1.0f : fmaf(tz, tz, fmaf(ty, ty, tx * tx));
            draw_count++;
            instance_count += ready_count;
        }
    }
    out->draw_count = draw_count;
    out->instance_count = instance_count;
}

void gvo_thread_scratch(uint32_t index, uint32_t records, GvoCullOut* out);
void gvo_scratch_lock(void);
void gvo_scratch_unlock(void);

typedef struct Task8 {
    const GvoSoa* s; const GvoMeshPool* mp; const GvoView* view; const GvoFrustum* fr; const GvoHiz* hiz;
    uint32_t lo, hi; GvoCullOut local; GvoCullOut* combined; _Atomic uint32_t* draw; _Atomic uint32_t* inst;
} Task8;

static void* task8_main(void* arg)
{
    Task8* t = (Task8*)arg;
    gvo_prepare_meshes_range_avx2(t->s, t->mp, t->view, t->fr, t->hiz, t->lo, t->hi, &t->local);
    const uint32_t off = atomic_fetch_add(t->draw, t->local.draw_count); /* mesh.cpp:177-183 */
    atomic_fetch_add(t->inst, t->local.instance_count);
    memcpy(t->combined->visible_idx + off, t->local.visible_idx, (size_t)t->local.draw_count * 4);
    memcpy(t->combined->baked_model + (size_t)off * 12, t->local.baked_model, (size_t)t->local.draw_count * 48);
    memcpy(t->combined->distance_sq + off, t->local.distance_sq, (size_t)t->local.draw_count * 4);
    return NULL;
}

void gvo_prepare_meshes_avx2(const GvoSoa* s, const GvoMeshPool* mp, const GvoView* view, const GvoHiz* hiz,
                             uint32_t threads, GvoCullOut* out)
{
    GvoFrustum fr;
    gvo_frustum_from_view_proj(view->view_proj, &fr);
    const uint32_t count = s->mesh_count;
    out->draw_count = out->instance_count = 0;
    if (!count)
        return;
    if (threads <= 1) {
        gvo_prepare_meshes_range_avx2(s, mp, view, &fr, hiz, 0, count, out);
        return;
    }
    const uint32_t task_count = count > threads ? threads : count; /* thread-pool.cpp:179-181 */
    uint32_t per = (uint32_t)ceilf((float)count / (float)task_count);
    Task8* tasks = (Task8*)calloc(task_count, sizeof(Task8));
    pthread_t* tids = (pthread_t*)calloc(task_count, sizeof(pthread_t));
    _Atomic uint32_t draw = 0, inst = 0;
    gvo_scratch_lock();
    for (uint32_t i = 0; i < task_count; i++) {
        Task8* t = &tasks[i];
        t->lo = per * i; t->hi = count < t->lo + per ? count : t->lo + per;
        if (t->lo >= t->hi) continue;
        const uint32_t n = t->hi - t->lo;
        t->s = s; t->mp = mp; t->view = view; t->fr = &fr; t->hiz = hiz; t->combined = out; t->draw = &draw; t->inst = &inst;
        gvo_thread_scratch(i, n, &t->local); /* threadMeshes[threadIndex]: persistent, grow-only  mesh.cpp:377-395 */
    }
    void** argv = (void**)malloc(sizeof(void*) * task_count);
    int argc = 0;
    for (uint32_t i = 0; i < task_count; i++) if (tasks[i].s) argv[argc++] = &tasks[i];
    gvo_pool_run(task8_main, argv, argc, (int)threads);
    free(argv);
    gvo_scratch_unlock();
    out->draw_count = atomic_load(&draw);
    out->instance_count = atomic_load(&inst);
    free(tasks); free(tids);
}
