/*
 * gv_oracle.c — scalar CPU restatement of the visibility hot path. TEST INFRASTRUCTURE ONLY
 * (see gv_oracle.h header: parity unpinned; who may link this).
 *
 * Build: gcc -O2 -march=haswell -ffp-contract=off -fno-fast-math (oracle/Makefile). Every
 * multiply that feeds an add is an explicit fmaf(), so the bits do not depend on the compiler's
 * contraction mode; the HIP kernels use the same written order.
 */
#include "gv_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * math — absent upstream (cfnptr/math); canonical order defined here (SURVEY.md §8c "must define").
 * ---------------------------------------------------------------------------------------------- */

/* math::calcModel(position, rotation, scale) call sites transform.hpp:199,207,224: T * R * S with R
 * from the (unit) quaternion. Column-major out, bottom row (0,0,0,1). */
void gvo_calc_model(const float pos[3], const float rot[4], const float scale[3], float out[16])
{
    const float x = rot[0], y = rot[1], z = rot[2], w = rot[3];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float zz = z * z2, yy = y * y2;
    const float wx = w * x2, wy = w * y2, wz = w * z2;
    const float r00 = 1.0f - fmaf(y, y2, zz);
    const float r11 = 1.0f - fmaf(x, x2, zz);
    const float r22 = 1.0f - fmaf(x, x2, yy);
    const float r10 = fmaf(x, y2, wz), r01 = fmaf(x, y2, -wz);
    const float r20 = fmaf(x, z2, -wy), r02 = fmaf(x, z2, wy);
    const float r21 = fmaf(y, z2, wx), r12 = fmaf(y, z2, -wx);
    const float sx = scale[0], sy = scale[1], sz = scale[2];
    out[0] = r00 * sx; out[1] = r10 * sx; out[2] = r20 * sx; out[3] = 0.0f;
    out[4] = r01 * sy; out[5] = r11 * sy; out[6] = r21 * sy; out[7] = 0.0f;
    out[8] = r02 * sz; out[9] = r12 * sz; out[10] = r22 * sz; out[11] = 0.0f;
    out[12] = pos[0]; out[13] = pos[1]; out[14] = pos[2]; out[15] = 1.0f;
}

/* f32x4x4 operator* (transform.hpp:209, graphics.cpp:243): out.c[j][i] = sum_k a.c[k][i] * b.c[j][k],
 * accumulated k = 0..3 as an fmaf chain starting from +0 — the order a v_mfma_f32_4x4x1 chain
 * with a zero C operand produces, and the order the VALU kernel writes. */
void gvo_mul4x4(const float a[16], const float b[16], float out[16])
{
    float r[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            float acc = 0.0f;
            for (int k = 0; k < 4; k++)
                acc = fmaf(a[k * 4 + i], b[j * 4 + k], acc);
            r[j * 4 + i] = acc;
        }
    memcpy(out, r, sizeof(r));
}

/* The chain product parentModel * model (transform.hpp:209). Model matrices are AFFINE: the bottom row is
 * (0,0,0,1) by construction (calcModel) and is carried as that constant, never recomputed — for finite inputs exactly
 * what the full 4x4 chain yields (0*x + ... + 1*1), for non-finite inputs it keeps 0*inf = NaN out of the bottom row.
 * Rows 0..2: the same fmaf chain as gvo_mul4x4 with b's bottom-row element taken as the constant 0 (columns 0..2) or
 * 1 (column 3). Every implementation (AVX2, VALU kernel, MFMA kernel) follows this definition. */
static void mul_affine(const float a[16], const float b[16], float out[16])
{
    float r[16];
    for (int j = 0; j < 4; j++) {
        const float b3 = j == 3 ? 1.0f : 0.0f;
        for (int i = 0; i < 3; i++) {
            float acc = fmaf(a[0 * 4 + i], b[j * 4 + 0], 0.0f);
            acc = fmaf(a[1 * 4 + i], b[j * 4 + 1], acc);
            acc = fmaf(a[2 * 4 + i], b[j * 4 + 2], acc);
            acc = fmaf(a[3 * 4 + i], b3, acc);
            r[j * 4 + i] = acc;
        }
        r[j * 4 + 3] = b3;
    }
    memcpy(out, r, sizeof(r));
}

/* Frustum(viewProj) (mesh.cpp:815,867,869,900,902): Gribb-Hartmann rows of a column-major matrix
 * for a [0,1] clip-space depth: left/right/bottom/top, z >= 0, z <= w. Planes normalised by
 * 1/sqrtf(|n|^2); planes with |n|^2 < 1e-12 are dropped (the z >= 0 plane of the infinite
 * reversed-Z projection, camera.hpp:115-116, degenerates to (0,0,0,near)). */
void gvo_frustum_from_view_proj(const float vp[16], GvoFrustum* out)
{
    float row[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            row[r][c] = vp[c * 4 + r];
    float p[6][4];
    for (int c = 0; c < 4; c++) {
        p[0][c] = row[3][c] + row[0][c];
        p[1][c] = row[3][c] - row[0][c];
        p[2][c] = row[3][c] + row[1][c];
        p[3][c] = row[3][c] - row[1][c];
        p[4][c] = row[2][c];
        p[5][c] = row[3][c] - row[2][c];
    }
    out->count = 0;
    memset(out->planes, 0, sizeof(out->planes));
    for (int i = 0; i < 6; i++) {
        const float len2 = fmaf(p[i][2], p[i][2], fmaf(p[i][1], p[i][1], p[i][0] * p[i][0]));
        if (!(len2 >= 1e-12f))
            continue;
        const float inv = 1.0f / sqrtf(len2);
        float* q = out->planes[out->count++];
        for (int c = 0; c < 4; c++)
            q[c] = p[i][c] * inv;
    }
}

/* 8 local-space corners (bit0 -> x, bit1 -> y, bit2 -> z select max) through the affine model. */
static void aabb_corners(const float mn[3], const float mx[3], const float m[16], float cx[8], float cy[8], float cz[8])
{
    for (int k = 0; k < 8; k++) {
        const float x = (k & 1) ? mx[0] : mn[0];
        const float y = (k & 2) ? mx[1] : mn[1];
        const float z = (k & 4) ? mx[2] : mn[2];
        cx[k] = fmaf(m[0], x, fmaf(m[4], y, fmaf(m[8], z, m[12])));
        cy[k] = fmaf(m[1], x, fmaf(m[5], y, fmaf(m[9], z, m[13])));
        cz[k] = fmaf(m[2], x, fmaf(m[6], y, fmaf(m[10], z, m[14])));
    }
}

/* isBehindFrustum(frustum, aabb, model) (mesh.hpp:145): behind iff some plane has all 8
 * transformed corners at signed distance < 0. */
int gvo_is_behind_frustum(const GvoFrustum* f, const float mn[3], const float mx[3], const float model[16])
{
    float cx[8], cy[8], cz[8];
    aabb_corners(mn, mx, model, cx, cy, cz);
    for (uint32_t p = 0; p < f->count; p++) {
        const float* n = f->planes[p];
        int all_behind = 1;
        for (int k = 0; k < 8; k++) {
            const float d = fmaf(n[0], cx[k], fmaf(n[1], cy[k], fmaf(n[2], cz[k], n[3])));
            if (!(d < 0.0f)) {
                all_behind = 0;
                break;
            }
        }
        if (all_behind)
            return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * transform chain — transform.hpp:197-214
 * ---------------------------------------------------------------------------------------------- */
static inline const uint8_t* tslot(const GvoTransformPool* tp, uint32_t slot)
{
    return tp->base + (size_t)slot * tp->stride;
}
static inline uint32_t lookup_transform(const GvoTransformPool* tp, uint32_t entity)
{
    if (entity == 0 || entity >= tp->entity_capacity)
        return GVO_NONE;
    return tp->entity_to_transform[entity];
}

void gvo_transform_calc_model(const GvoTransformPool* tp, uint32_t slot, const float cam[3], float out[16])
{
    const uint8_t* t = tslot(tp, slot);
    float model[16];
    /* auto model = math::calcModel(posChildCount, rotation, scaleChildCap);  (:199) */
    gvo_calc_model((const float*)(t + tp->off_position), (const float*)(t + tp->off_rotation),
                   (const float*)(t + tp->off_scale), model);
    if (*(t + tp->off_model_with_ancestors)) { /* :200 */
        uint32_t next_parent;
        memcpy(&next_parent, t + tp->off_parent, 4);
        while (next_parent) { /* :204 */
            const uint32_t ps = lookup_transform(tp, next_parent); /* manager->get<TransformComponent> :206 */
            if (ps == GVO_NONE)
                break; /* reference would throw; a dangling parent ends the chain here */
            const uint8_t* p = tslot(tp, ps);
            float parent_model[16];
            gvo_calc_model((const float*)(p + tp->off_position), (const float*)(p + tp->off_rotation),
                           (const float*)(p + tp->off_scale), parent_model);
            mul_affine(parent_model, model, model); /* model = parentModel * model;  :209 */
            memcpy(&next_parent, p + tp->off_parent, 4);
        }
    }
    /* math::translate(-cameraPosition, model)  (:211,:213): pre-translation of an affine matrix */
    model[12] = model[12] - cam[0];
    model[13] = model[13] - cam[1];
    model[14] = model[14] - cam[2];
    memcpy(out, model, sizeof(model));
}

void gvo_world_matrices(const GvoTransformPool* tp, uint32_t first, uint32_t count, float* out12)
{
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    for (uint32_t s = 0; s < count; s++) {
        float m[16];
        uint32_t entity;
        memcpy(&entity, tslot(tp, first + s) + tp->off_entity, 4);
        if (!entity) {
            memset(out12 + (size_t)s * 12, 0, 48);
            continue;
        }
        gvo_transform_calc_model(tp, first + s, zero, m);
        for (int c = 0; c < 4; c++)
            memcpy(out12 + (size_t)s * 12 + c * 3, m + c * 4, 12);
    }
}

/* the same, slot range split over `threads` threads like ThreadPool::addItems (thread-pool.cpp:173-200) */
typedef struct WorldTask {
    const GvoTransformPool* tp;
    uint32_t first, count;
    float* out12;
} WorldTask;
static void* world_task_main(void* arg)
{
    const WorldTask* t = (const WorldTask*)arg;
    gvo_world_matrices(t->tp, t->first, t->count, t->out12);
    return NULL;
}
void gvo_pool_run(void* (*fn)(void*), void** args, int count, int threads);
void gvo_world_matrices_mt(const GvoTransformPool* tp, uint32_t first, uint32_t count, float* out12, uint32_t threads)
{
    if (threads <= 1 || count < 4096u) {
        gvo_world_matrices(tp, first, count, out12);
        return;
    }
    const uint32_t task_count = count > threads ? threads : count;
    const uint32_t per = (uint32_t)ceilf((float)count / (float)task_count);
    WorldTask* tasks = (WorldTask*)calloc(task_count, sizeof(WorldTask));
    void** argv = (void**)calloc(task_count, sizeof(void*));
    int argc = 0;
    for (uint32_t i = 0; i < task_count; i++) {
        const uint32_t lo = per * i, hi = count < lo + per ? count : lo + per;
        if (lo >= hi)
            continue;
        tasks[argc] = (WorldTask){tp, first + lo, hi - lo, out12 + (size_t)lo * 12};
        argv[argc] = &tasks[argc];
        argc++;
    }
    gvo_pool_run(world_task_main, argv, argc, (int)threads);
    free(argv);
    free(tasks);
}

/* ------------------------------------------------------------------------------------------------
 * Hi-Z pyramid — shaders/hiz.frag:23-63, source/system/render/hiz.cpp:24-57
 * ---------------------------------------------------------------------------------------------- */
uint32_t gvo_calc_mip_count(uint32_t w, uint32_t h)
{
    uint32_t m = w > h ? w : h, n = 0;
    while (m) {
        n++;
        m >>= 1;
    }
    return n; /* floor(log2(max)) + 1 */
}

uint64_t gvo_hiz_layout(uint32_t w, uint32_t h, GvoHiz* out)
{
    out->width = w;
    out->height = h;
    out->mip_count = gvo_calc_mip_count(w, h);
    uint64_t off = 0;
    uint32_t cw = w, ch = h;
    for (uint32_t k = 0; k < out->mip_count && k < 16; k++) {
        out->mip_w[k] = cw;
        out->mip_h[k] = ch;
        out->mip_offset[k] = off;
        if (k >= 1)
            off += (uint64_t)cw * ch;
        cw = cw / 2 > 1 ? cw / 2 : 1; /* frameSize = max(frameSize / 2u, uint2::one)  hiz.cpp:55 */
        ch = ch / 2 > 1 ? ch / 2 : 1;
    }
    return off;
}

/* RG16F pyramid (HizRenderSystem::bufferFormat = SfloatR16G16, include/garden/system/render/hiz.hpp:41). The reference lets
 * the render target round float -> half (nearest, implementation-defined), which is not conservative; the build's variant
 *  rounds min toward -inf and max toward +inf, so a stored pair still bounds every covered depth. These
 * two functions ARE the format: binary16 with subnormals, NaN kept (quiet), overflow to +-inf / +-65504 by direction. */
uint16_t gvo_half_directed(float f, int up)
{
    uint32_t bits;
    memcpy(&bits, &f, 4);
    const uint32_t sign = bits >> 31, mag = bits & 0x7FFFFFFFu;
    const uint32_t grow = (uint32_t)(up != 0) != sign; /* the direction increases the magnitude */
    uint32_t h;
    if (mag > 0x7F800000u)
        h = 0x7E00u; /* NaN */
    else if (mag == 0x7F800000u)
        h = 0x7C00u;
    else if (mag >= 0x47800000u) /* >= 2^16: beyond the largest half */
        h = grow ? 0x7C00u : 0x7BFFu;
    else {
        const int e = (int)(mag >> 23) - 127;
        if (e >= -14) { /* normal half; a carry out of the mantissa moves into the exponent (up to inf) */
            const uint32_t mant = mag & 0x7FFFFFu;
            h = ((uint32_t)(e + 15) << 10) | (mant >> 13);
            if ((mant & 0x1FFFu) && grow)
                h++;
        } else { /* subnormal half (or zero): units of 2^-24 */
            const uint32_t mant = (mag >> 23) ? ((mag & 0x7FFFFFu) | 0x800000u) : 0u; /* float subnormals: < 2^-126, all remainder */
            const int shift = 13 + (-14 - e); /* >= 14 */
            uint32_t rem;
            if (shift > 31 || mant == 0) {
                h = 0;
                rem = mag;
            } else {
                h = mant >> shift;
                rem = mant & ((1u << shift) - 1u);
            }
            if (rem && grow)
                h++;
        }
    }
    return (uint16_t)((sign << 15) | h);
}

float gvo_half_to_float(uint16_t hb)
{
    const uint32_t sign = (uint32_t)(hb >> 15) << 31, e = (hb >> 10) & 31u, m = hb & 0x3FFu;
    uint32_t bits;
    if (e == 31u)
        bits = sign | 0x7F800000u | (m << 13);
    else if (e != 0u)
        bits = sign | ((e + 112u) << 23) | (m << 13);
    else if (m == 0u)
        bits = sign;
    else { /* subnormal: m * 2^-24, normalised */
        uint32_t mm = m, ee = 113u;
        while (!(mm & 0x400u)) {
            mm <<= 1;
            ee--;
        }
        bits = sign | (ee << 23) | ((mm & 0x3FFu) << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline void src_texel(const GvoHiz* hz, const float* mips, uint32_t level, uint32_t x, uint32_t y, float* mn, float* mxv)
{
    if (level == 0) { /* HIZ_VARIANT_FIRST: minMax = (d, d)  hiz.frag:57-60 */
        const float d = hz->depth[(size_t)y * hz->width + x];
        *mn = d;
        *mxv = d;
    } else {
        const float* t = mips + 2 * (hz->mip_offset[level] + (uint64_t)y * hz->mip_w[level] + x);
        *mn = t[0];
        *mxv = t[1];
    }
}

/* rows [py0, py1) of level k from level k - 1 (hiz.frag:23-63) */
static void hiz_build_rows(const GvoHiz* hz, float* mips, int rule_and_format, uint32_t k, uint32_t py0, uint32_t py1)
{
    const int rule = rule_and_format & 0xFF, rg16f = (rule_and_format & GVO_HIZ_FORMAT_RG16F) != 0;
    const uint32_t sw = hz->mip_w[k - 1], sh = hz->mip_h[k - 1];
    const uint32_t dw = hz->mip_w[k];
    const int odd_x = (sw & 1u) != 0, odd_y = (sh & 1u) != 0; /* isPrevLevelOdd  hiz.frag:35 */
    for (uint32_t py = py0; py < py1; py++)
        for (uint32_t px = 0; px < dw; px++) {
            /* textureGather footprint of the 2x2 quad at 2p (hiz.frag:29-33); clamp-to-edge
             * sampler semantics for a 1-texel-wide source. */
            const uint32_t x0 = 2 * px, y0 = 2 * py;
            const uint32_t x1 = x0 + 1 < sw ? x0 + 1 : sw - 1, y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
            const uint32_t x2 = x0 + 2 < sw ? x0 + 2 : sw - 1, y2 = y0 + 2 < sh ? y0 + 2 : sh - 1;
            float mn, mx, a, b;
            src_texel(hz, mips, k - 1, x0, y0, &mn, &mx);
#define ACC(X, Y) do { src_texel(hz, mips, k - 1, (X), (Y), &a, &b); mn = a < mn ? a : mn; mx = b > mx ? b : mx; } while (0)
            ACC(x1, y0);
            ACC(x0, y1);
            ACC(x1, y1);
            if (odd_x) { /* hiz.frag:36-41: gatherOffset(1,0) .y .z = column 2p.x+2, rows 2p.y+1, 2p.y */
                ACC(x2, y1);
                ACC(x2, y0);
                if (odd_y) /* hiz.frag:43-47: texel (2p + 2) */
                    ACC(x2, y2);
            }
            if (odd_y) {
                /* hiz.frag:49-55: gatherOffset(0,1) components .y .z = texels (2p.x+1, 2p.y+2) and
                 * (2p.x+1, 2p.y+1): the reference as written does NOT read (2p.x, 2p.y+2). */
                ACC(x1, y2);
                ACC(x1, y1);
                if (rule == GVO_HIZ_RULE_CONSERVATIVE)
                    ACC(x0, y2); /* the full extra row, so that min/max bound every covered texel */
            }
#undef ACC
            if (rg16f) { /* what an RG16F texel holds (levels >= 2 reduce stored halfs: already exact) */
                mn = gvo_half_to_float(gvo_half_directed(mn, 0));
                mx = gvo_half_to_float(gvo_half_directed(mx, 1));
            }
            float* d = mips + 2 * (hz->mip_offset[k] + (uint64_t)py * dw + px);
            d[0] = mn;
            d[1] = mx;
        }
}

void gvo_hiz_build(GvoHiz* hz, float* mips, int rule)
{
    hz->mips = mips;
    for (uint32_t k = 1; k < hz->mip_count; k++)
        hiz_build_rows(hz, mips, rule, k, 0, hz->mip_h[k]);
}

/* The same pyramid with every level's rows split over `threads` threads like ThreadPool::addItems
 * (thread-pool.cpp:173-200); levels stay sequential (one render pass per mip, hiz.cpp:148-164). Same bits: every
 * destination texel is computed by exactly one thread from the finished level below. */
typedef struct HizRowsTask {
    const GvoHiz* hz;
    float* mips;
    int rule;
    uint32_t k, py0, py1;
} HizRowsTask;
static void* hiz_rows_main(void* arg)
{
    const HizRowsTask* t = (const HizRowsTask*)arg;
    hiz_build_rows(t->hz, t->mips, t->rule, t->k, t->py0, t->py1);
    return NULL;
}
void gvo_pool_run(void* (*fn)(void*), void** args, int count, int threads);
void gvo_hiz_build_mt(GvoHiz* hz, float* mips, int rule, uint32_t threads)
{
    hz->mips = mips;
    if (threads < 1)
        threads = 1;
    HizRowsTask* tasks = (HizRowsTask*)calloc(threads, sizeof(HizRowsTask));
    void** argv = (void**)calloc(threads, sizeof(void*));
    for (uint32_t k = 1; k < hz->mip_count; k++) {
        const uint32_t rows = hz->mip_h[k];
        if (threads == 1 || (uint64_t)rows * hz->mip_w[k] < 16384u) { /* not worth a fan-out */
            hiz_build_rows(hz, mips, rule, k, 0, rows);
            continue;
        }
        const uint32_t task_count = rows > threads ? threads : rows;
        const uint32_t per = (uint32_t)ceilf((float)rows / (float)task_count);
        int argc = 0;
        for (uint32_t i = 0; i < task_count; i++) {
            const uint32_t lo = per * i, hi = rows < lo + per ? rows : lo + per;
            if (lo >= hi)
                continue;
            tasks[argc] = (HizRowsTask){hz, mips, rule, k, lo, hi};
            argv[argc] = &tasks[argc];
            argc++;
        }
        gvo_pool_run(hiz_rows_main, argv, argc, (int)threads);
    }
    free(argv);
    free(tasks);
}

/* Build-defined occlusion query, SURVEY.md §8a-7' (no reference: SURVEY.md F3). */
int gvo_hiz_occluded(const GvoHiz* hz, const float vp[16], const float mn[3], const float mx[3], const float model[16])
{
    float cx[8], cy[8], cz[8];
    aabb_corners(mn, mx, model, cx, cy, cz);
    float umin = 0, umax = 0, vmin = 0, vmax = 0, znear = 0;
    for (int k = 0; k < 8; k++) {
        const float clx = fmaf(vp[0], cx[k], fmaf(vp[4], cy[k], fmaf(vp[8], cz[k], vp[12])));
        const float cly = fmaf(vp[1], cx[k], fmaf(vp[5], cy[k], fmaf(vp[9], cz[k], vp[13])));
        const float clz = fmaf(vp[2], cx[k], fmaf(vp[6], cy[k], fmaf(vp[10], cz[k], vp[14])));
        const float clw = fmaf(vp[3], cx[k], fmaf(vp[7], cy[k], fmaf(vp[11], cz[k], vp[15])));
        if (!(clw > 0.0f))
            return 0; /* touches/crosses the camera plane: cannot bound -> visible */
        const float rcp = 1.0f / clw;
        const float u = fmaf(clx * rcp, 0.5f, 0.5f);
        const float v = fmaf(cly * rcp, 0.5f, 0.5f);
        const float zc = clz * rcp;
        if (k == 0) {
            umin = umax = u;
            vmin = vmax = v;
            znear = zc;
        } else { /* IEEE minNum/maxNum (a NaN operand is ignored), the semantics of v_min_f32/v_max_f32 */
            umin = fminf(umin, u);
            umax = fmaxf(umax, u);
            vmin = fminf(vmin, v);
            vmax = fmaxf(vmax, v);
            znear = fmaxf(znear, zc);
        }
    }
#define CLAMP01(a) ((a) > 0.0f ? ((a) < 1.0f ? (a) : 1.0f) : 0.0f)
    umin = CLAMP01(umin);
    umax = CLAMP01(umax);
    vmin = CLAMP01(vmin);
    vmax = CLAMP01(vmax);
#undef CLAMP01
    const int W = (int)hz->width, H = (int)hz->height;
    int ix0 = (int)(umin * (float)W), ix1 = (int)(umax * (float)W);
    int iy0 = (int)(vmin * (float)H), iy1 = (int)(vmax * (float)H);
    ix0 = ix0 < W - 1 ? ix0 : W - 1;
    ix1 = ix1 < W - 1 ? ix1 : W - 1;
    iy0 = iy0 < H - 1 ? iy0 : H - 1;
    iy1 = iy1 < H - 1 ? iy1 : H - 1;
    /* smallest level at which the pixel rect touches <= 2x2 texels */
    uint32_t level = 0;
    while (level + 1 < hz->mip_count &&
           (((ix1 >> level) - (ix0 >> level)) > 1 || ((iy1 >> level) - (iy0 >> level)) > 1))
        level++;
    const int lw = (int)hz->mip_w[level], lh = (int)hz->mip_h[level];
    int tx0 = ix0 >> level, tx1 = ix1 >> level, ty0 = iy0 >> level, ty1 = iy1 >> level;
    tx0 = tx0 < lw - 1 ? tx0 : lw - 1;
    tx1 = tx1 < lw - 1 ? tx1 : lw - 1;
    ty0 = ty0 < lh - 1 ? ty0 : lh - 1;
    ty1 = ty1 < lh - 1 ? ty1 : lh - 1;
    float zfar, a, b;
    src_texel(hz, hz->mips, level, (uint32_t)tx0, (uint32_t)ty0, &zfar, &b);
    src_texel(hz, hz->mips, level, (uint32_t)tx1, (uint32_t)ty0, &a, &b);
    zfar = a < zfar ? a : zfar;
    src_texel(hz, hz->mips, level, (uint32_t)tx0, (uint32_t)ty1, &a, &b);
    zfar = a < zfar ? a : zfar;
    src_texel(hz, hz->mips, level, (uint32_t)tx1, (uint32_t)ty1, &a, &b);
    zfar = a < zfar ? a : zfar;
    /* reversed-Z (depth.gsl:20-21): larger = nearer. Occluded iff the box's nearest depth is
     * behind the farthest occluder depth over its footprint; ties are visible. */
    return znear < zfar ? 1 : 0;
}

/* ------------------------------------------------------------------------------------------------
 * the hot loop — source/system/render/mesh.cpp:111-184 (and sorted twin :187-262)
 * ---------------------------------------------------------------------------------------------- */
void gvo_prepare_meshes_range(const GvoMeshPool* mp, const GvoTransformPool* tp, const GvoView* view,
                              const GvoFrustum* frustum, const GvoHiz* hiz, uint32_t item_offset,
                              uint32_t item_end, GvoCullOut* out)
{
    const int is_not_shadow_pass = view->shadow_pass < 0; /* mesh.cpp:121 */
    uint32_t draw_count = 0, instance_count = 0;
    for (uint32_t i = item_offset; i < item_end; i++) { /* :137 */
        uint8_t* mesh = mp->base + (size_t)i * mp->stride; /* :139 */
        const float* amin = (const float*)(mesh + mp->off_aabb_min);
        const float* amax = (const float*)(mesh + mp->off_aabb_max);
        /* aabb.getSize(); fixW()  (:140): w never vetoes */
        const float sx = amax[0] - amin[0], sy = amax[1] - amin[1], sz = amax[2] - amin[2];
        uint32_t entity;
        memcpy(&entity, mesh + mp->off_entity, 4);
        if (!entity || !*(mesh + mp->off_is_enabled) || (sx <= 0.0f && sy <= 0.0f && sz <= 0.0f)) { /* :142 */
            if (is_not_shadow_pass)
                *(mesh + mp->off_is_visible) = 0; /* :144 */
            continue;
        }
        const uint32_t ts = lookup_transform(tp, entity); /* manager->tryGet<TransformComponent>  :149 */
        int active = 0;
        if (ts != GVO_NONE) {
            const uint8_t* t = tslot(tp, ts);
            active = *(t + tp->off_self_active) && *(t + tp->off_ancestors_active); /* isActive()  transform.hpp:110 */
        }
        if (!active) { /* :150 */
            if (is_not_shadow_pass)
                *(mesh + mp->off_is_visible) = 0;
            continue;
        }
        float model[16];
        gvo_transform_calc_model(tp, ts, view->camera_position, model); /* :157 */
        /* getReadyMeshesAsync default predicate (mesh.hpp:142-146) */
        uint32_t ready_count = gvo_is_behind_frustum(frustum, amin, amax, model) ? 0u : 1u; /* :158 */
        /* build-defined occlusion stage (no reference): only after frustum survival */
        if (ready_count && view->use_hiz && hiz && gvo_hiz_occluded(hiz, view->view_proj, amin, amax, model))
            ready_count = 0;
        if (ready_count && mp->ready_base) { /* derived predicate: `return descriptorSet ? 1 : 0` sprite.cpp:96 */
            const uint8_t* r = mp->ready_base + (size_t)i * mp->ready_stride;
            if (mp->ready_width == 4)
                memcpy(&ready_count, r, 4);
            else
                ready_count = *r;
        }
        if (ready_count == 0) { /* :159 */
            if (is_not_shadow_pass)
                *(mesh + mp->off_is_visible) = 0;
            continue;
        }
        if (is_not_shadow_pass)
            *(mesh + mp->off_is_visible) = 1; /* :166 */
        out->visible_idx[draw_count] = i; /* componentOffset = i * componentSize  :170 */
        float* bm = out->baked_model + (size_t)draw_count * 12;
        for (int c = 0; c < 4; c++)
            memcpy(bm + c * 3, model + c * 4, 12); /* (float4x3)bakedModel  :171 */
        const float tx = model[12] + view->camera_offset[0];
        const float ty = model[13] + view->camera_offset[1];
        const float tz = model[14] + view->camera_offset[2];
        /* :172 lengthSq3(getTranslation(model) + cameraOffset); sorted twin :250-251 */
        out->distance_sq[draw_count] = view->distance_2d ? model[14] + 1.0f : fmaf(tz, tz, fmaf(ty, ty, tx * tx));
        draw_count++;
        instance_count += ready_count; /* :174 */
    }
    out->draw_count = draw_count;
    out->instance_count = instance_count;
}

/* Persistent worker pool (the reference's foreground ThreadPool is persistent too: source/thread-pool.cpp:56-110;
 * the calling thread takes part, thread-pool.cpp:203-215). One global pool, grown on demand. */
typedef struct GvoThreadPool {
    pthread_t* tids;
    int workers;
    pthread_mutex_t mu;
    pthread_cond_t work_cv, done_cv;
    void* (*fn)(void*);
    void** args;
    int next, total, pending;
    unsigned generation;
} GvoThreadPool;
static GvoThreadPool g_pool = {NULL, 0, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER,
                               NULL, NULL, 0, 0, 0, 0};
static pthread_mutex_t g_pool_run = PTHREAD_MUTEX_INITIALIZER;

static void* pool_worker(void* unused)
{
    (void)unused;
    unsigned seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.generation == seen)
            pthread_cond_wait(&g_pool.work_cv, &g_pool.mu);
        seen = g_pool.generation;
        while (g_pool.next < g_pool.total) {
            const int i = g_pool.next++;
            pthread_mutex_unlock(&g_pool.mu);
            g_pool.fn(g_pool.args[i]);
            pthread_mutex_lock(&g_pool.mu);
            if (--g_pool.pending == 0)
                pthread_cond_signal(&g_pool.done_cv);
        }
    }
    return NULL;
}

/* runs fn(args[i]) for i in [0, count) on up to `threads` threads including the caller */
void gvo_pool_run(void* (*fn)(void*), void** args, int count, int threads)
{
    pthread_mutex_lock(&g_pool_run);
    const int want = threads - 1;
    if (want > g_pool.workers) {
        g_pool.tids = (pthread_t*)realloc(g_pool.tids, sizeof(pthread_t) * (size_t)want);
        for (int i = g_pool.workers; i < want; i++) {
            pthread_create(&g_pool.tids[i], NULL, pool_worker, NULL);
            pthread_detach(g_pool.tids[i]);
        }
        g_pool.workers = want;
    }
    pthread_mutex_lock(&g_pool.mu);
    g_pool.fn = fn;
    g_pool.args = args;
    g_pool.total = count;
    g_pool.next = 0;
    g_pool.pending = count;
    g_pool.generation++;
    pthread_cond_broadcast(&g_pool.work_cv);
    while (g_pool.next < g_pool.total) {
        const int i = g_pool.next++;
        pthread_mutex_unlock(&g_pool.mu);
        fn(args[i]);
        pthread_mutex_lock(&g_pool.mu);
        --g_pool.pending;
    }
    while (g_pool.pending > 0)
        pthread_cond_wait(&g_pool.done_cv, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_unlock(&g_pool_run);
}

/* threadMeshes[threadIndex] (mesh.cpp:127-131): per-task record scratch that persists from frame to frame and only
 * grows (mesh.cpp:377-395), so a frame costs no allocation and no fresh-page faults. Not re-entrant across concurrent
 * gvo_prepare_meshes* calls (neither is the reference's MeshRenderSystem). */
typedef struct Scratch {
    uint32_t* idx;
    float* model;
    float* dist;
    uint32_t cap;
} Scratch;
static Scratch* g_scratch = NULL;
static uint32_t g_scratch_count = 0;
static pthread_mutex_t g_scratch_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_mutex_t g_scratch_owner = PTHREAD_MUTEX_INITIALIZER;
/* one threaded prepare at a time owns the scratch (lock before the first gvo_thread_scratch, unlock after the run) */
void gvo_scratch_lock(void) { pthread_mutex_lock(&g_scratch_owner); }
void gvo_scratch_unlock(void) { pthread_mutex_unlock(&g_scratch_owner); }
void gvo_thread_scratch(uint32_t index, uint32_t records, GvoCullOut* out)
{
    pthread_mutex_lock(&g_scratch_mu);
    if (index >= g_scratch_count) {
        const uint32_t want = index + 1;
        g_scratch = (Scratch*)realloc(g_scratch, sizeof(Scratch) * want);
        memset(g_scratch + g_scratch_count, 0, sizeof(Scratch) * (want - g_scratch_count));
        g_scratch_count = want;
    }
    Scratch* s = &g_scratch[index];
    if (records > s->cap) {
        free(s->idx);
        free(s->model);
        free(s->dist);
        s->idx = (uint32_t*)malloc((size_t)records * 4);
        s->model = (float*)malloc((size_t)records * 48);
        s->dist = (float*)malloc((size_t)records * 4);
        s->cap = records;
    }
    out->visible_idx = s->idx;
    out->baked_model = s->model;
    out->distance_sq = s->dist;
    pthread_mutex_unlock(&g_scratch_mu);
}

typedef struct RangeTask {
    const GvoMeshPool* mp;
    const GvoTransformPool* tp;
    const GvoView* view;
    const GvoFrustum* frustum;
    const GvoHiz* hiz;
    uint32_t item_offset, item_end;
    GvoCullOut local;           /* threadMeshes[threadIndex]  mesh.cpp:127-131 */
    GvoCullOut* combined;       /* combinedMeshes */
    _Atomic uint32_t* draw_count;
    _Atomic uint32_t* instance_count;
} RangeTask;

static void* range_task_main(void* arg)
{
    RangeTask* t = (RangeTask*)arg;
    gvo_prepare_meshes_range(t->mp, t->tp, t->view, t->frustum, t->hiz, t->item_offset, t->item_end, &t->local);
    /* drawCount.fetch_add; instanceCount.fetch_add; memcpy into combinedMeshes  mesh.cpp:177-183 */
    const uint32_t draw_offset = atomic_fetch_add(t->draw_count, t->local.draw_count);
    atomic_fetch_add(t->instance_count, t->local.instance_count);
    memcpy(t->combined->visible_idx + draw_offset, t->local.visible_idx, (size_t)t->local.draw_count * 4);
    memcpy(t->combined->baked_model + (size_t)draw_offset * 12, t->local.baked_model, (size_t)t->local.draw_count * 48);
    memcpy(t->combined->distance_sq + draw_offset, t->local.distance_sq, (size_t)t->local.draw_count * 4);
    return NULL;
}

void gvo_prepare_meshes(const GvoMeshPool* mp, const GvoTransformPool* tp, const GvoView* view,
                        const GvoHiz* hiz, uint32_t threads, GvoCullOut* out)
{
    GvoFrustum frustum;
    gvo_frustum_from_view_proj(view->view_proj, &frustum); /* Frustum(cc.viewProj)  mesh.cpp:900 */
    const uint32_t count = mp->occupancy;
    out->draw_count = out->instance_count = 0;
    if (count == 0)
        return;
    if (threads <= 1) { /* !useThreading: write straight into combinedMeshes  mesh.cpp:133-136 */
        gvo_prepare_meshes_range(mp, tp, view, &frustum, hiz, 0, count, out);
        return;
    }
    /* ThreadPool::addItems  thread-pool.cpp:173-200 */
    const uint32_t task_count = count > threads ? threads : count;
    const uint32_t count_per_thread = (uint32_t)ceilf((float)count / (float)task_count);
    RangeTask* tasks = (RangeTask*)calloc(task_count, sizeof(RangeTask));
    pthread_t* tids = (pthread_t*)calloc(task_count, sizeof(pthread_t));
    _Atomic uint32_t draw_count = 0, instance_count = 0;
    uint32_t launched = 0;
    gvo_scratch_lock();
    for (uint32_t i = 0; i < task_count; i++) {
        RangeTask* t = &tasks[i];
        t->item_offset = count_per_thread * i;
        t->item_end = count < t->item_offset + count_per_thread ? count : t->item_offset + count_per_thread;
        if (t->item_offset >= t->item_end)
            continue;
        const uint32_t n = t->item_end - t->item_offset;
        t->mp = mp; t->tp = tp; t->view = view; t->frustum = &frustum; t->hiz = hiz;
        t->combined = out; t->draw_count = &draw_count; t->instance_count = &instance_count;
        gvo_thread_scratch(i, n, &t->local);
        launched = i + 1;
    }
    /* foreground pool: the calling thread participates (ThreadPool::wait  thread-pool.cpp:203-215) */
    void** argv = (void**)malloc(sizeof(void*) * (launched ? launched : 1));
    int argc = 0;
    for (uint32_t i = 0; i < launched; i++)
        if (tasks[i].mp)
            argv[argc++] = &tasks[i];
    gvo_pool_run(range_task_main, argv, argc, (int)threads);
    free(argv);
    gvo_scratch_unlock();
    out->draw_count = atomic_load(&draw_count);
    out->instance_count = atomic_load(&instance_count);
    free(tasks);
    free(tids);
}

/* sortMeshes  mesh.cpp:265-328: std::sort by distanceSq (operator< mesh.hpp:196,204). Ties broken
 * by index here so the oracle's order is reproducible (std::sort is unstable in the reference). */
typedef struct SortKey { float d; uint32_t idx; uint32_t src; } SortKey;
static int cmp_asc(const void* a, const void* b)
{
    const SortKey* x = (const SortKey*)a; const SortKey* y = (const SortKey*)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static int cmp_desc(const void* a, const void* b)
{
    const SortKey* x = (const SortKey*)a; const SortKey* y = (const SortKey*)b;
    if (x->d > y->d) return -1;
    if (x->d < y->d) return 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
void gvo_sort_records(GvoCullOut* out, int descending)
{
    const uint32_t n = out->draw_count;
    if (n < 2)
        return;
    SortKey* keys = (SortKey*)malloc((size_t)n * sizeof(SortKey));
    for (uint32_t i = 0; i < n; i++) {
        keys[i].d = out->distance_sq[i];
        keys[i].idx = out->visible_idx[i];
        keys[i].src = i;
    }
    qsort(keys, n, sizeof(SortKey), descending ? cmp_desc : cmp_asc);
    uint32_t* idx = (uint32_t*)malloc((size_t)n * 4);
    float* bm = (float*)malloc((size_t)n * 48);
    float* ds = (float*)malloc((size_t)n * 4);
    for (uint32_t i = 0; i < n; i++) {
        idx[i] = keys[i].idx;
        ds[i] = keys[i].d;
        memcpy(bm + (size_t)i * 12, out->baked_model + (size_t)keys[i].src * 12, 48);
    }
    memcpy(out->visible_idx, idx, (size_t)n * 4);
    memcpy(out->baked_model, bm, (size_t)n * 48);
    memcpy(out->distance_sq, ds, (size_t)n * 4);
    free(keys); free(idx); free(bm); free(ds);
}
