/*
 * gv_census.cpp — PARITY-RISK CENSUS (test / analysis infrastructure, like everything in oracle/; never linked into
 * the product). The arithmetic of this path is build-defined because cfnptr/math is absent (gv_oracle.h: "parity
 * unpinned"): this file re-evaluates the per-entity visibility decision of prepareUnsortedMeshes (mesh.cpp:137-175)
 * under the OTHER operation orders a real cfnptr/math build could legally have, and counts the entities whose
 * decision differs from the canonical one — turning "unpinned" into a number.
 *
 *   variant 0  canonical   = gv_oracle.c (must reproduce its decisions exactly; checked by the caller)
 *   variant 1  float64     every product, sum and comparison in double (inputs are the same float32 values)
 *   variant 2  association the chain as ((M_root * M_p1) * M_p2) * M_self — root first ("world[parent] * local", what a
 *                          level-ordered propagation computes) instead of transform.hpp:209's child-first
 *                          parentModel * model; canonical arithmetic otherwise
 *   variant 3  unfused     the source form `a*b + c*d + e*f + g` with every product and sum rounded separately
 *                          (-ffp-contract=off on the un-fused source: clang's default for C++, MSVC /fp:precise)
 *   variant 4  contracted  the same source form as GCC contracts it (-ffp-contract=fast, GCC's default; the build
 *                          compiles with -march=haswell, cmake/compile-options.cmake:34-46): left to right, each next
 *                          product fused into the running sum, the trailing constant added last
 * The Hi-Z query itself is build-defined (the reference has none) and has no alternative to census: variants hand their
 * own model matrix to the canonical query (variant 1: its double model rounded to float).
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {
#include "gv_oracle.h"
}

namespace {

struct Canonical {
    typedef float T;
    static T m2(T a, T b, T c) { return fmaf(a, b, c); }  // a*b + c
    static T p3(T c0, T x, T c1, T y, T c2, T z, T c3) { return fmaf(c0, x, fmaf(c1, y, fmaf(c2, z, c3))); }
    static T p4(T a0, T b0, T a1, T b1, T a2, T b2, T a3, T b3) { return fmaf(a3, b3, fmaf(a2, b2, fmaf(a1, b1, fmaf(a0, b0, 0.0f)))); }
};
struct Float64 {
    typedef double T;
    static T m2(T a, T b, T c) { return a * b + c; }
    static T p3(T c0, T x, T c1, T y, T c2, T z, T c3) { return c0 * x + c1 * y + c2 * z + c3; }
    static T p4(T a0, T b0, T a1, T b1, T a2, T b2, T a3, T b3) { return a0 * b0 + a1 * b1 + a2 * b2 + a3 * b3; }
};
// volatile stores keep the compiler from contracting or re-associating (this file is also built with -ffp-contract=off)
static inline float rnd(float v)
{
    volatile float t = v;
    return t;
}
struct Unfused {
    typedef float T;
    static T m2(T a, T b, T c) { return rnd(rnd(a * b) + c); }
    static T p3(T c0, T x, T c1, T y, T c2, T z, T c3) { return rnd(rnd(rnd(rnd(c0 * x) + rnd(c1 * y)) + rnd(c2 * z)) + c3); }
    static T p4(T a0, T b0, T a1, T b1, T a2, T b2, T a3, T b3)
    {
        return rnd(rnd(rnd(rnd(a0 * b0) + rnd(a1 * b1)) + rnd(a2 * b2)) + rnd(a3 * b3));
    }
};
struct Contracted {
    typedef float T;
    static T m2(T a, T b, T c) { return fmaf(a, b, c); }
    static T p3(T c0, T x, T c1, T y, T c2, T z, T c3) { return rnd(fmaf(c2, z, fmaf(c1, y, rnd(c0 * x))) + c3); }
    static T p4(T a0, T b0, T a1, T b1, T a2, T b2, T a3, T b3) { return fmaf(a3, b3, fmaf(a2, b2, fmaf(a1, b1, rnd(a0 * b0)))); }
};

template <class A>
struct Mat {  // affine, column-major columns c0..c3, rows 0..2
    typename A::T c[4][3];
};

template <class A>
Mat<A> calc_model(const float* pos, const float* rot, const float* scl)
{
    typedef typename A::T T;
    const T x = rot[0], y = rot[1], z = rot[2], w = rot[3];
    const T x2 = x + x, y2 = y + y, z2 = z + z;
    const T zz = z * z2, yy = y * y2;
    const T wx = w * x2, wy = w * y2, wz = w * z2;
    const T r00 = T(1) - A::m2(y, y2, zz), r11 = T(1) - A::m2(x, x2, zz), r22 = T(1) - A::m2(x, x2, yy);
    const T r10 = A::m2(x, y2, wz), r01 = A::m2(x, y2, -wz);
    const T r20 = A::m2(x, z2, -wy), r02 = A::m2(x, z2, wy);
    const T r21 = A::m2(y, z2, wx), r12 = A::m2(y, z2, -wx);
    const T sx = scl[0], sy = scl[1], sz = scl[2];
    Mat<A> m;
    m.c[0][0] = r00 * sx; m.c[0][1] = r10 * sx; m.c[0][2] = r20 * sx;
    m.c[1][0] = r01 * sy; m.c[1][1] = r11 * sy; m.c[1][2] = r21 * sy;
    m.c[2][0] = r02 * sz; m.c[2][1] = r12 * sz; m.c[2][2] = r22 * sz;
    m.c[3][0] = pos[0]; m.c[3][1] = pos[1]; m.c[3][2] = pos[2];
    return m;
}

template <class A>
Mat<A> mul(const Mat<A>& a, const Mat<A>& b)  // a * b, bottom rows (0,0,0,1)
{
    typedef typename A::T T;
    Mat<A> r;
    for (int j = 0; j < 4; j++) {
        const T b3 = j == 3 ? T(1) : T(0);
        for (int i = 0; i < 3; i++)
            r.c[j][i] = A::p4(a.c[0][i], b.c[j][0], a.c[1][i], b.c[j][1], a.c[2][i], b.c[j][2], a.c[3][i], b3);
    }
    return r;
}

inline uint32_t lookup(const GvoTransformPool* tp, uint32_t entity)
{
    if (entity == 0 || entity >= tp->entity_capacity)
        return GVO_NONE;
    return tp->entity_to_transform[entity];
}

template <class A>
Mat<A> slot_model(const GvoTransformPool* tp, uint32_t slot)
{
    const uint8_t* t = tp->base + (size_t)slot * tp->stride;
    return calc_model<A>((const float*)(t + tp->off_position), (const float*)(t + tp->off_rotation), (const float*)(t + tp->off_scale));
}

// TransformComponent::calcModel(cameraPosition), transform.hpp:197-214; root_first = the other association
template <class A>
Mat<A> chain_model(const GvoTransformPool* tp, uint32_t slot, const float cam[3], bool root_first)
{
    const uint8_t* t = tp->base + (size_t)slot * tp->stride;
    Mat<A> model = slot_model<A>(tp, slot);
    if (*(t + tp->off_model_with_ancestors)) {
        uint32_t chain[64];
        int depth = 0;
        uint32_t next_parent;
        memcpy(&next_parent, t + tp->off_parent, 4);
        while (next_parent && depth < 64) {
            const uint32_t ps = lookup(tp, next_parent);
            if (ps == GVO_NONE)
                break;
            chain[depth++] = ps;
            memcpy(&next_parent, tp->base + (size_t)ps * tp->stride + tp->off_parent, 4);
        }
        if (!root_first) {
            for (int k = 0; k < depth; k++)
                model = mul<A>(slot_model<A>(tp, chain[k]), model);  // model = parentModel * model
        } else if (depth) {
            Mat<A> acc = slot_model<A>(tp, chain[depth - 1]);  // the root
            for (int k = depth - 2; k >= 0; k--)
                acc = mul<A>(acc, slot_model<A>(tp, chain[k]));  // world[parent] = world[grandparent] * local[parent]
            model = mul<A>(acc, model);
        }
    }
    for (int i = 0; i < 3; i++)
        model.c[3][i] = model.c[3][i] - (typename A::T)cam[i];
    return model;
}

// behind iff some plane has all 8 corners at distance < 0; *margin = min over planes of |max corner distance|:
// how far the decision is from flipping (world units)
template <class A>
bool behind_frustum(const GvoFrustum* f, const float* mn, const float* mx, const Mat<A>& m, double* margin)
{
    typedef typename A::T T;
    T cx[8], cy[8], cz[8];
    for (int k = 0; k < 8; k++) {
        const T x = (k & 1) ? mx[0] : mn[0], y = (k & 2) ? mx[1] : mn[1], z = (k & 4) ? mx[2] : mn[2];
        cx[k] = A::p3(m.c[0][0], x, m.c[1][0], y, m.c[2][0], z, m.c[3][0]);
        cy[k] = A::p3(m.c[0][1], x, m.c[1][1], y, m.c[2][1], z, m.c[3][1]);
        cz[k] = A::p3(m.c[0][2], x, m.c[1][2], y, m.c[2][2], z, m.c[3][2]);
    }
    bool behind = false;
    double best = 1e300;
    for (uint32_t p = 0; p < f->count; p++) {
        const float* n = f->planes[p];
        T worst = A::p3((T)n[0], cx[0], (T)n[1], cy[0], (T)n[2], cz[0], (T)n[3]);
        bool all_behind = worst < T(0);
        for (int k = 1; k < 8; k++) {
            const T d = A::p3((T)n[0], cx[k], (T)n[1], cy[k], (T)n[2], cz[k], (T)n[3]);
            all_behind = all_behind && (d < T(0));
            if (d > worst || worst != worst)
                worst = d;
        }
        behind = behind || all_behind;
        const double a = std::fabs((double)worst);
        if (a < best)
            best = a;
    }
    *margin = best;
    return behind;
}

template <class A>
void to_float16(const Mat<A>& m, float out[16])
{
    for (int j = 0; j < 4; j++) {
        for (int i = 0; i < 3; i++)
            out[j * 4 + i] = (float)m.c[j][i];
        out[j * 4 + 3] = j == 3 ? 1.0f : 0.0f;
    }
}

// decision: 0 = filtered out (no arithmetic involved), 1 = rejected by the frustum, 2 = occluded (Hi-Z), 3 = visible
template <class A>
void run_range(const GvoMeshPool* mp, const GvoTransformPool* tp, const GvoView* view, const GvoFrustum* fr, const GvoHiz* hiz,
               bool root_first, uint32_t lo, uint32_t hi, uint8_t* decision, float* margin)
{
    for (uint32_t i = lo; i < hi; i++) {
        const uint8_t* mesh = mp->base + (size_t)i * mp->stride;
        decision[i] = 0;
        margin[i] = 0.0f;
        uint32_t entity;
        memcpy(&entity, mesh + mp->off_entity, 4);
        if (!entity || !*(mesh + mp->off_is_enabled))
            continue;
        const float* mn = (const float*)(mesh + mp->off_aabb_min);
        const float* mx = (const float*)(mesh + mp->off_aabb_max);
        if (mx[0] - mn[0] <= 0.0f && mx[1] - mn[1] <= 0.0f && mx[2] - mn[2] <= 0.0f)
            continue;
        const uint32_t ts = lookup(tp, entity);
        if (ts == GVO_NONE)
            continue;
        const uint8_t* t = tp->base + (size_t)ts * tp->stride;
        if (!(*(t + tp->off_self_active) && *(t + tp->off_ancestors_active)))
            continue;
        const Mat<A> model = chain_model<A>(tp, ts, view->camera_position, root_first);
        double m = 0;
        if (behind_frustum<A>(fr, mn, mx, model, &m)) {
            decision[i] = 1;
        } else {
            decision[i] = 3;
            if (view->use_hiz && hiz) {
                float model16[16];
                to_float16<A>(model, model16);
                if (gvo_hiz_occluded(hiz, view->view_proj, mn, mx, model16))
                    decision[i] = 2;
            }
        }
        margin[i] = (float)m;
    }
}

}  // namespace

extern "C" void gvo_census(const GvoMeshPool* mp, const GvoTransformPool* tp, const GvoView* view, const GvoHiz* hiz, int variant,
                           uint32_t threads, uint8_t* decision, float* margin)
{
    GvoFrustum fr;
    gvo_frustum_from_view_proj(view->view_proj, &fr);
    const uint32_t n = mp->occupancy;
    if (threads < 1)
        threads = 1;
    std::vector<std::thread> pool;
    const uint32_t per = (n + threads - 1) / threads;
    for (uint32_t k = 0; k < threads; k++) {
        const uint32_t lo = std::min(n, per * k), hi = std::min(n, per * (k + 1));
        if (lo >= hi)
            continue;
        pool.emplace_back([=, &fr] {
            switch (variant) {
            case 1: run_range<Float64>(mp, tp, view, &fr, hiz, false, lo, hi, decision, margin); break;
            case 2: run_range<Canonical>(mp, tp, view, &fr, hiz, true, lo, hi, decision, margin); break;
            case 3: run_range<Unfused>(mp, tp, view, &fr, hiz, false, lo, hi, decision, margin); break;
            case 4: run_range<Contracted>(mp, tp, view, &fr, hiz, false, lo, hi, decision, margin); break;
            default: run_range<Canonical>(mp, tp, view, &fr, hiz, false, lo, hi, decision, margin); break;
            }
        });
    }
    for (auto& th : pool)
        th.join();
}
