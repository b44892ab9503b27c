// tk_bvh_device.h — ray queries against the flattened scene BVH: closest hit
// and occlusion, with the per-lane traversal stack in LDS.
//
// The CPU renderer walks a two-level BVH (objects, then one BVH per mesh), far
// child first and without culling boxes against the current hit
// (src/BVH.cpp:245-288, src/Bounds3.hpp:96-125).  Here one BVH covers every
// triangle and sphere, the near child is visited first and boxes beyond the
// current hit are skipped; the closest hit is the same because every
// primitive test and the strict `t < t_max` update rule are kept
// (src/Triangle.cpp:15-52, src/Sphere.cpp:14-55).
#pragma once
#include "tk_device.h"

namespace tkd {

constexpr int kStackDepth = 32;      // per-lane entries; the builder bounds the tree depth
constexpr int kMaxLeaf = 2;          // primitives per BVH leaf
constexpr int kDeviceBuildMin = 4096; // scenes from this many primitives on are built on the GPU by default
constexpr int kTraceBlock = 256;     // threads per block of every kernel that traces

// LDS stack: entry e of thread t lives at stack[e * STRIDE + t] (STRIDE = block
// size) so that the 64 lanes of a wave touch 64 consecutive banks.
template <int STRIDE>
struct LdsStackT {
    int* base;
    TKD void push(int sp, int v) { base[sp * STRIDE] = v; }
    TKD int pop(int sp) { return base[sp * STRIDE]; }
};
using LdsStack = LdsStackT<kTraceBlock>;

struct Hit {
    int prim;        // original primitive id, -1 = miss
    double t;        // r.t_max after the query
    double u, v;     // barycentrics of a triangle hit
};

struct SurfacePoint {
    F3 p, perr, n, wo;
    int mat;         // material index
    int light;       // light index or -1
};

// Interval arithmetic of src/EFloat.h:8-144 for the sphere quadratic.
struct EF {
    float v, lo, hi;
};
TKD EF ef_make(float v) { EF r; r.v = v; r.lo = v; r.hi = v; return r; }
TKD EF ef_make_err(float v, float err) {
    EF r; r.v = v;
    if (err == 0.0f) { r.lo = v; r.hi = v; }
    else { r.lo = next_down(v - err); r.hi = next_up(v + err); }
    return r;
}
TKD EF ef_add(EF a, EF b) { EF r; r.v = a.v + b.v; r.lo = next_down(a.lo + b.lo); r.hi = next_up(a.hi + b.hi); return r; }
TKD EF ef_sub(EF a, EF b) { EF r; r.v = a.v - b.v; r.lo = next_down(a.lo - b.hi); r.hi = next_up(a.hi - b.lo); return r; }
TKD float std_min(float a, float b) { return b < a ? b : a; }
TKD float std_max(float a, float b) { return a < b ? b : a; }
TKD EF ef_mul(EF a, EF b) {
    EF r; r.v = a.v * b.v;
    float p0 = a.lo * b.lo, p1 = a.hi * b.lo, p2 = a.lo * b.hi, p3 = a.hi * b.hi;
    r.lo = next_down(std_min(std_min(p0, p1), std_min(p2, p3)));
    r.hi = next_up(std_max(std_max(p0, p1), std_max(p2, p3)));
    return r;
}
TKD EF ef_div(EF a, EF b) {
    EF r; r.v = a.v / b.v;
    if (b.lo < 0 && b.hi > 0) { r.lo = -INFINITY; r.hi = INFINITY; }
    else {
        float d0 = a.lo / b.lo, d1 = a.hi / b.lo, d2 = a.lo / b.hi, d3 = a.hi / b.hi;
        r.lo = next_down(std_min(std_min(d0, d1), std_min(d2, d3)));
        r.hi = next_up(std_max(std_max(d0, d1), std_max(d2, d3)));
    }
    return r;
}
TKD bool ef_quadratic(EF a, EF b, EF c, EF* t0, EF* t1) {      // src/EFloat.h:125-143
    float discrim = b.v * b.v - 4.0f * a.v * c.v;
    if (discrim < 0) return false;
    float root = sqrtf(discrim);
    EF froot = ef_make_err(root, kEps * root);
    EF q;
    if (b.v < 0) q = ef_mul(ef_sub(b, froot), ef_make(-0.5f));
    else q = ef_mul(ef_add(b, froot), ef_make(-0.5f));
    *t0 = ef_div(q, a);
    *t1 = ef_div(c, q);
    if (t0->v > t1->v) { EF t = *t0; *t0 = *t1; *t1 = t; }
    return true;
}

// Sphere::intersect up to the choice of the root (src/Sphere.cpp:14-37).  Kept
// out of line: scenes hold a handful of spheres at most, and the interval
// arithmetic is several hundred instructions that would otherwise be repeated
// inside every traversal loop.
TKD bool sphere_hit_t(float4 sph, F3 o, F3 d, double tmax, float* thit) {
    EF ox = ef_sub(ef_make(o.x), ef_make(sph.x));
    EF oy = ef_sub(ef_make(o.y), ef_make(sph.y));
    EF oz = ef_sub(ef_make(o.z), ef_make(sph.z));
    EF dx = ef_make(d.x), dy = ef_make(d.y), dz = ef_make(d.z);
    EF a = ef_add(ef_add(ef_mul(dx, dx), ef_mul(dy, dy)), ef_mul(dz, dz));
    EF b = ef_mul(ef_add(ef_add(ef_mul(dx, ox), ef_mul(dy, oy)), ef_mul(dz, oz)), ef_make(2.0f));
    EF rr = ef_make(sph.w);
    EF c = ef_sub(ef_add(ef_add(ef_mul(ox, ox), ef_mul(oy, oy)), ef_mul(oz, oz)), ef_mul(rr, rr));
    EF t0, t1;
    if (!ef_quadratic(a, b, c, &t0, &t1)) return false;
    if ((double)t0.hi > tmax || t1.lo <= 0) return false;
    EF hit = t0;
    if (hit.lo <= 0) {
        hit = t1;
        if ((double)hit.hi > tmax) return false;
    }
    *thit = hit.v;
    return true;
}

// Triangle::intersect (src/Triangle.cpp:15-37): fp32 products, fp64 scalars.
// The fp64 division only runs for rays that pass a conservative fp32 screen:
// u = un/det and v = vn/det can only lie in [0,1] (and t be positive) when the
// numerators have the sign of det and do not exceed it by more than rounding,
// so anything the screen drops is also dropped by the exact test below.
TKD bool tri_hit(const TriIsect& T, F3 o, F3 d, double tmax, double* tout, double* uout, double* vout) {
    F3 e1 = ld3(T.e1), e2 = ld3(T.e2);
    F3 pvec = cross3(d, e2);
    float detf = dot3(e1, pvec);
    float adet = fabsf(detf);
    if (adet < kEps) return false;
    F3 tvec = sub3(o, ld3(T.v0));
    float unf = dot3(tvec, pvec);
    float sgn = detf < 0.f ? -1.f : 1.f;
    float us = unf * sgn;                       // u * |det|
    const float slack = adet * 1.0e-5f;
    if (us < 0.f || us > adet + slack) return false;
    F3 qvec = cross3(tvec, e1);
    float vnf = dot3(d, qvec);
    float vs = vnf * sgn;                       // v * |det|
    if (vs < 0.f || us + vs > adet + slack) return false;
    float tnf = dot3(e2, qvec);
    if (tnf * sgn <= 0.f) return false;         // t <= 0
    double det = (double)detf;
    double det_inv = 1. / det;
    double u = (double)unf * det_inv;
    if (u < 0 || u > 1) return false;
    double v = (double)vnf * det_inv;
    if (v < 0 || u + v > 1) return false;
    double t = (double)tnf * det_inv;
    if (t > 0 && t < tmax) { *tout = t; *uout = u; *vout = v; return true; }
    return false;
}

// Slab test with the tMax padding of src/Bounds3.hpp:108-118; returns the
// entry distance.  For a direction without zero components (finite inverses)
// this min / max form decides exactly as the CPU code does: each bound goes
// through (bound - origin) * inverse, two roundings, as it does there.
//
// The arithmetic runs on pairs of floats: the six bounds of a box lie in memory
// as (min.x min.y) (min.z max.x) (max.y max.z), so with the ray's origin and
// inverse direction held as the pairs (x y) (z x) (y z) each subtraction and
// each product serves two bounds with the same two roundings per component.
typedef float F2 __attribute__((ext_vector_type(2)));
struct RayPairs {
    F2 o_xy, o_zx, o_yz, i_xy, i_zx, i_yz;
};
TKD RayPairs ray_pairs(F3 o, F3 dinv) {
    RayPairs r;
    r.o_xy = F2{o.x, o.y}; r.o_zx = F2{o.z, o.x}; r.o_yz = F2{o.y, o.z};
    r.i_xy = F2{dinv.x, dinv.y}; r.i_zx = F2{dinv.z, dinv.x}; r.i_yz = F2{dinv.y, dinv.z};
    return r;
}
TKD bool box_hit_pairs(F2 p0, F2 p1, F2 p2, const RayPairs& r, float tmax, float* tnear) {
    const float pad = 1 + 2 * gamma_n(3);
    const F2 a = (p0 - r.o_xy) * r.i_xy;      // tx0 ty0
    const F2 b = (p1 - r.o_zx) * r.i_zx;      // tz0 tx1
    const F2 c = (p2 - r.o_yz) * r.i_yz;      // ty1 tz1
    float tn = fmaxf(fmaxf(fminf(a.x, b.y), fminf(a.y, c.x)), fminf(b.x, c.y));
    float tf = fminf(fminf(fmaxf(a.x, b.y), fmaxf(a.y, c.x)), fmaxf(b.x, c.y)) * pad;
    *tnear = tn;
    return tn <= tf && tf > 0 && tn < tmax;
}
// The same test for a ray with an infinite inverse (a zero direction component),
// statement by statement as Bounds3::IntersectP makes it (src/Bounds3.hpp:96-125):
// the slab bound is picked by `dir > 0`, so a +0 component takes the far plane
// as entry (+inf: the ray misses every box), a -0 component passes, and a NaN
// from 0 * inf goes through the comparisons as it does there.
TKD bool box_hit_zero_dir(const float* bmin, const float* bmax, F3 o, F3 d, F3 dinv, float tmax, float* tnear) {
    const float pad = 1 + 2 * gamma_n(3);
    const bool px = d.x > 0, py = d.y > 0, pz = d.z > 0;
    float tMin = ((px ? bmin[0] : bmax[0]) - o.x) * dinv.x;
    float tMax = ((px ? bmax[0] : bmin[0]) - o.x) * dinv.x;
    float tyMin = ((py ? bmin[1] : bmax[1]) - o.y) * dinv.y;
    float tyMax = ((py ? bmax[1] : bmin[1]) - o.y) * dinv.y;
    tMax *= pad;
    tyMax *= pad;
    *tnear = 0.f;
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = ((pz ? bmin[2] : bmax[2]) - o.z) * dinv.z;
    float tzMax = ((pz ? bmax[2] : bmin[2]) - o.z) * dinv.z;
    tzMax *= pad;
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    *tnear = tMin;
    return (tMin <= tMax) && (tMax > 0) && !(tMin >= tmax);
}

TKD void prim_test(const DScene& sc, const TriIsect& T, F3 o, F3 d, Hit* h) {
    if (T.prim < sc.n_tris) {
        double t, u, v;
        if (tri_hit(T, o, d, h->t, &t, &u, &v)) { h->t = t; h->u = u; h->v = v; h->prim = T.prim; }
    } else {
        float th;
        if (sphere_hit_t(sc.spheres[T.prim - sc.n_tris], o, d, h->t, &th)) { h->t = (double)th; h->prim = T.prim; }
    }
}

// A leaf holds at most kMaxLeaf primitives; the records of the first two are
// fetched together (independent loads in flight) before either is tested.
TKD void leaf_test(const DScene& sc, int enc, F3 o, F3 d, Hit* h, int* prims_tested = nullptr) {
    const int first = enc >> 3, count = (enc & 7) + 1;
    if (prims_tested) *prims_tested += count;
    const TriIsect T0 = sc.tri_isect[first];
    const TriIsect T1 = sc.tri_isect[first + (count > 1 ? 1 : 0)];
    prim_test(sc, T0, o, d, h);
    if (count > 1) prim_test(sc, T1, o, d, h);
    if (kMaxLeaf > 2) {
        for (int i = 2; i < count; ++i) {
            const TriIsect T = sc.tri_isect[first + i];
            prim_test(sc, T, o, d, h);
        }
    }
}

// Closest hit, or occlusion when any_hit is set (stops at the first hit below
// tmax, enough for the boolean Scene::intersectP answers,
// src/core/Scene.cpp:27-31), as a walk that can be left and taken up again.
//
// Two nested loops.  The inner one only walks inner nodes (two slab tests and
// a push); a lane leaves it when it holds a leaf, and the loop ends as soon as
// half of the wave's rays wait with a leaf (kLeafShare): waiting for every lane
// would leave most of them idle behind the few with a long descent, and testing
// each leaf at once would run the primitive tests with a few lanes at a time.
// The lanes still descending sit out the leaf tests and carry on afterwards.
// Stack entries are inner nodes (>= 0) or leaves (< 0, ~encoding); the per-lane
// stack in LDS carries over between calls.
//
// A wave that traces one ray per lane to the end (trace_ray) runs as long as
// its slowest ray.  The kernels that walk whole paths call walk_rays with
// `rest` > 0 instead: it returns once all but `rest` rays of the wave have
// finished, those lanes go on with their paths (and start their next rays),
// and the slow rays simply continue in the next call.
constexpr int kWalkOver = 0x7fffffff;   // RayWalk::cur of a lane whose ray has finished (or that has none)
#ifndef TK_LEAF_SHARE
#define TK_LEAF_SHARE 8                 // (a build flag for sweeps)
#endif
constexpr int kLeafShare = TK_LEAF_SHARE;   // sixteenths of the rays in flight that must hold a leaf before leaves are tested
// What a counting instance of a walk adds up per lane: nodes of the tree fetched (a node holds the boxes of
// both children; a node of the eight-children tree counts as four, by its size) and primitives tested.
// node_turns / leaf_turns: turns of the inner-node loop and of the leaf phase this lane sat through, node_busy:
// those of the node loop in which it still had a ray under way (walking or waiting with a leaf), leaf_lanes:
// leaf phases in which it tested a leaf — what share of the lanes each section of the walk keeps busy.
struct WalkCount {
    int nodes, prims;
    int node_turns, node_busy, leaf_turns, leaf_lanes;
};
struct RayWalk {
    int cur;        // node to visit next, ~leaf to test next, or kWalkOver
    int sp;
    float tcull;
    F3 dinv;
    bool zero_dir;  // some component of the direction has an infinite inverse: the slow exact slab test
    Hit h;
};
TKD void walk_begin(RayWalk& w, const DScene& sc, F3 d, double tmax) {
    w.h.prim = -1; w.h.t = tmax; w.h.u = 0; w.h.v = 0;
    // Ray ctor: direction_inv = float(1.0 / double(dir)) (src/Ray.hpp:21).  The correctly rounded
    // fp32 quotient is the same number: 1/x for a 24-bit x never lies within a double's rounding
    // error of a midpoint between two floats.
    w.dinv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    w.zero_dir = isinf(w.dinv.x) || isinf(w.dinv.y) || isinf(w.dinv.z);
    w.tcull = tmax >= 3.0e38 ? 3.4e38f : (float)tmax * 1.0000002f;
    w.sp = 0;
    w.cur = sc.n_nodes == 0 ? kWalkOver : 0;
}
// Advances the rays in flight (lanes with cur != kWalkOver) until at most
// `rest` of them are unfinished.  Called by all the lanes that take part, in
// step; the votes below count those lanes only.
// The whole walk of a ray with an infinite inverse (a zero direction component), lane by lane with the exact
// slab test.  Such rays are rare (axis-parallel mirror chains), so they are walked to the end where they are met
// and the loops of walk_rays_t carry one slab test only.
template <bool COUNT, class STACK>
TKD void walk_zero_dir(const DScene& sc, RayWalk& w, F3 o, F3 d, bool any_hit, STACK st, WalkCount* wc) {
    while (w.cur != kWalkOver) {
        if (w.cur >= 0) {
            const BvhNode nd = sc.nodes[w.cur];
            if (COUNT) ++wc->nodes;
            float tl, tr;
            const bool hl = box_hit_zero_dir(nd.lmin, nd.lmax, o, d, w.dinv, w.tcull, &tl);
            const bool hr = box_hit_zero_dir(nd.rmin, nd.rmax, o, d, w.dinv, w.tcull, &tr);
            if (hl && hr) {
                const bool right_first = tr < tl;
                st.push(w.sp, right_first ? nd.left : nd.right);
                ++w.sp;
                w.cur = right_first ? nd.right : nd.left;
            } else if (hl) {
                w.cur = nd.left;
            } else if (hr) {
                w.cur = nd.right;
            } else if (w.sp > 0) {
                --w.sp;
                w.cur = st.pop(w.sp);
            } else {
                w.cur = kWalkOver;
            }
        } else {
            leaf_test(sc, ~w.cur, o, d, &w.h, COUNT ? &wc->prims : nullptr);
            w.tcull = w.h.t >= 3.0e38 ? 3.4e38f : (float)w.h.t * 1.0000002f;
            if ((any_hit && w.h.prim >= 0) || w.sp == 0) w.cur = kWalkOver;
            else { --w.sp; w.cur = st.pop(w.sp); }
        }
    }
}
template <bool COUNT, class STACK>
TKD void walk_rays_t(const DScene& sc, RayWalk& w, F3 o, F3 d, bool any_hit, STACK st, int rest, WalkCount* wc) {
    if (__builtin_expect(w.zero_dir && w.cur != kWalkOver, 0)) walk_zero_dir<COUNT>(sc, w, o, d, any_hit, st, wc);
    const RayPairs rp = ray_pairs(o, w.dinv);
    while (true) {
        const int busy = __popcll(__ballot(w.cur != kWalkOver));
        // (the votes make the loop condition the same for the whole wave: a scalar branch)
        bool inner = (unsigned)w.cur < (unsigned)kWalkOver;
        int n_inner = __popcll(__ballot(inner));
        while (n_inner != 0 && (busy - n_inner) * 16 < busy * kLeafShare) {
            if (COUNT) { ++wc->node_turns; if (w.cur != kWalkOver) ++wc->node_busy; }
            if (inner) {
                const BvhNode nd = sc.nodes[w.cur];
                if (COUNT) ++wc->nodes;
                float tl, tr;
                const bool hl = box_hit_pairs(F2{nd.lmin[0], nd.lmin[1]}, F2{nd.lmin[2], nd.lmax[0]}, F2{nd.lmax[1], nd.lmax[2]}, rp, w.tcull, &tl);
                const bool hr = box_hit_pairs(F2{nd.rmin[0], nd.rmin[1]}, F2{nd.rmin[2], nd.rmax[0]}, F2{nd.rmax[1], nd.rmax[2]}, rp, w.tcull, &tr);
                if (hl && hr) {
                    const bool right_first = tr < tl;
                    st.push(w.sp, right_first ? nd.left : nd.right);
                    ++w.sp;
                    w.cur = right_first ? nd.right : nd.left;
                } else if (hl) {
                    w.cur = nd.left;
                } else if (hr) {
                    w.cur = nd.right;
                } else if (w.sp > 0) {
                    --w.sp;
                    w.cur = st.pop(w.sp);
                } else {
                    w.cur = kWalkOver;
                }
            }
            inner = (unsigned)w.cur < (unsigned)kWalkOver;
            n_inner = __popcll(__ballot(inner));
        }
        if (COUNT && __any(w.cur < 0)) ++wc->leaf_turns;
        if (w.cur < 0) {
            if (COUNT) ++wc->leaf_lanes;
            leaf_test(sc, ~w.cur, o, d, &w.h, COUNT ? &wc->prims : nullptr);
            w.tcull = w.h.t >= 3.0e38 ? 3.4e38f : (float)w.h.t * 1.0000002f;
            if ((any_hit && w.h.prim >= 0) || w.sp == 0) w.cur = kWalkOver;
            else { --w.sp; w.cur = st.pop(w.sp); }
        }
        if (__popcll(__ballot(w.cur != kWalkOver)) <= rest) break;
    }
}
template <class STACK>
TKD void walk_rays(const DScene& sc, RayWalk& w, F3 o, F3 d, bool any_hit, STACK st, int rest) {
    walk_rays_t<false>(sc, w, o, d, any_hit, st, rest, nullptr);
}
template <bool COUNT, class STACK>
TKD Hit trace_ray_t(const DScene& sc, F3 o, F3 d, double tmax, bool any_hit, STACK st, WalkCount* wc) {
    RayWalk w;
    walk_begin(w, sc, d, tmax);
    walk_rays_t<COUNT>(sc, w, o, d, any_hit, st, 0, wc);
    return w.h;
}
template <class STACK>
TKD Hit trace_ray(const DScene& sc, F3 o, F3 d, double tmax, bool any_hit, STACK st) {
    return trace_ray_t<false>(sc, o, d, tmax, any_hit, st, nullptr);
}

// ---------------------------------------------------------------- wide walk
// Closest hit (or occlusion) for up to a few rays of a wave, eight lanes per ray, in the tree of wide
// nodes: the eight lanes of a group test the eight child boxes of a node side by side, the leaves among
// the children that are hit are tested at once (each by the lane that holds it), the nearest inner child
// is visited next and the others go on the ray's stack with their entry distances.  A ray takes a third
// of the dependent round trips of the one-lane walk above, which is what a wave with few rays left waits
// for.  The boxes and the primitive tests are the same as in walk_rays and a box that holds a hit is never
// culled in either, so the hit is the same (up to the order in which primitives at exactly the same
// distance are met).
//
// Called by all 64 lanes; `want` marks the lanes with a ray (any number: eight are served at a time).
// The per-lane LDS stacks must be free (no one-lane walk in flight): a group keeps its stack in the
// eight columns of its lanes, references in the upper half of the rows, distances in the lower.
constexpr int kWideStack = kStackDepth / 2 * 8;   // entries a ray can hold
TKD uint32_t wide_order_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
template <bool COUNT, int STRIDE>
TKD Hit walk_wide_t(const DScene& sc, bool want, F3 o_in, F3 d_in, double tmax_in, bool any_hit_in, LdsStackT<STRIDE> st,
                    WalkCount* wc) {
    const int lane = (int)(threadIdx.x & 63);
    const int sub = lane & 7, grp = lane >> 3, lane0 = grp << 3;
    int* const col = st.base - lane + lane0;         // row 0 of the group's first column
    Hit result;
    result.prim = -1; result.t = tmax_in; result.u = 0; result.v = 0;
    unsigned long long todo = __ballot(want);
    while (todo) {
        // this batch: the lowest eight lanes with a ray; group g serves the g-th of them
        unsigned long long later = todo;
        for (int i = 0; i < 8 && later; ++i) later &= later - 1;
        const unsigned long long batch = todo & ~later;
        todo = later;
        unsigned long long m = batch;
        for (int i = 0; i < grp; ++i) m &= m - 1;
        bool active = m != 0;
        const int owner = active ? __ffsll((long long)m) - 1 : lane;
        const F3 o = mk3(__shfl(o_in.x, owner), __shfl(o_in.y, owner), __shfl(o_in.z, owner));
        const F3 d = mk3(__shfl(d_in.x, owner), __shfl(d_in.y, owner), __shfl(d_in.z, owner));
        const double tmax = __shfl(tmax_in, owner);
        const bool any_hit = __shfl(any_hit_in ? 1 : 0, owner) != 0;
        const F3 dinv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const bool zero_dir = isinf(dinv.x) || isinf(dinv.y) || isinf(dinv.z);
        const RayPairs rp = ray_pairs(o, dinv);
        float tcull = tmax >= 3.0e38 ? 3.4e38f : (float)tmax * 1.0000002f;
        Hit best;
        best.prim = -1; best.t = tmax; best.u = 0; best.v = 0;
        int cur = 0, sp = 0;
        while (__any(active)) {
            if (active) {
                const WideChild ch = sc.wide[cur].c[sub];
                if (COUNT && sub == 0) wc->nodes += 4;     // counted with the lanes that serve the ray, not with its owner
                float tn = 0.f;
                bool hit = false;
                if (ch.ref != kWideEmpty)
                    hit = zero_dir ? box_hit_zero_dir(ch.lo, ch.hi, o, d, dinv, tcull, &tn)
                                   : box_hit_pairs(F2{ch.lo[0], ch.lo[1]}, F2{ch.lo[2], ch.hi[0]}, F2{ch.hi[1], ch.hi[2]}, rp, tcull, &tn);
                Hit mine = best;
                mine.prim = -1;
                if (hit && ch.ref < 0) leaf_test(sc, ~ch.ref, o, d, &mine, COUNT ? &wc->prims : nullptr);
                // hits of this visit nearer than the best so far, lane by lane
                unsigned fm = (unsigned)(__ballot(mine.prim >= 0) >> lane0) & 0xffu;
                while (fm) {
                    const int l = lane0 + __ffs((int)fm) - 1;
                    fm &= fm - 1;
                    const double t = __shfl(mine.t, l), u = __shfl(mine.u, l), v = __shfl(mine.v, l);
                    const int prim = __shfl(mine.prim, l);
                    if (t < best.t) { best.t = t; best.u = u; best.v = v; best.prim = prim; }
                }
                tcull = best.t >= 3.0e38 ? 3.4e38f : (float)best.t * 1.0000002f;
                if (any_hit && best.prim >= 0) {
                    active = false;
                } else {
                    const bool in = hit && ch.ref >= 0 && tn < tcull;
                    const unsigned im = (unsigned)(__ballot(in) >> lane0) & 0xffu;
                    bool got = false;
                    if (im) {
                        uint32_t key = in ? ((wide_order_key(tn) & ~7u) | (uint32_t)sub) : 0xffffffffu;
                        key = min(key, (uint32_t)__shfl_xor((int)key, 1));
                        key = min(key, (uint32_t)__shfl_xor((int)key, 2));
                        key = min(key, (uint32_t)__shfl_xor((int)key, 4));
                        const int near = (int)(key & 7u);
                        const unsigned others = im & ~(1u << near);
                        if (in && sub != near) {
                            const int e = sp + __popc(others & ((1u << sub) - 1u));
                            col[(e >> 3) * STRIDE + (e & 7)] = ch.ref;
                            col[((e >> 3) + kStackDepth / 2) * STRIDE + (e & 7)] = __float_as_int(tn);
                        }
                        sp += __popc(others);
                        cur = __shfl(ch.ref, lane0 + near);
                        got = true;
                    }
                    while (!got && sp > 0) {      // the last entry that the best hit has not overtaken
                        --sp;
                        const int ref = col[(sp >> 3) * STRIDE + (sp & 7)];
                        const float tn_e = __int_as_float(col[((sp >> 3) + kStackDepth / 2) * STRIDE + (sp & 7)]);
                        if (tn_e < tcull) { cur = ref; got = true; }
                    }
                    if (!got) active = false;
                }
            }
        }
        // results to the lanes the rays came from: the r-th lane of the batch reads group r
        const bool mine_now = (batch >> lane) & 1ull;
        const int src = __popcll(batch & ((1ull << lane) - 1ull)) << 3;
        const double t = __shfl(best.t, src & 63), u = __shfl(best.u, src & 63), v = __shfl(best.v, src & 63);
        const int prim = __shfl(best.prim, src & 63);
        if (mine_now) { result.t = t; result.u = u; result.v = v; result.prim = prim; }
    }
    return result;
}
template <int STRIDE>
TKD Hit walk_wide(const DScene& sc, bool want, F3 o_in, F3 d_in, double tmax_in, bool any_hit_in, LdsStackT<STRIDE> st) {
    return walk_wide_t<false>(sc, want, o_in, d_in, tmax_in, any_hit_in, st, nullptr);
}

// Fills the surface record of a hit: Triangle::intersect :39-48 or
// Sphere::intersect :39-53, and the object's material / light.
TKD SurfacePoint surface_at(const DScene& sc, const Hit& h, F3 o, F3 d) {
    SurfacePoint s;
    if (h.prim < sc.n_tris) {
        const TriShade T = sc.tri_shade[h.prim];
        double u = h.u, v = h.v, w = 1 - u - v;
        float xs = (float)(fabs(w * T.v0[0]) + fabs(u * T.v1[0]) + fabs(v * T.v2[0]));
        float ys = (float)(fabs(w * T.v0[1]) + fabs(u * T.v1[1]) + fabs(v * T.v2[1]));
        float zs = (float)(fabs(w * T.v0[2]) + fabs(u * T.v1[2]) + fabs(v * T.v2[2]));
        s.perr = scale3(mk3(xs, ys, zs), gamma_n(7));
        s.p = add3(add3(scale3(ld3(T.v0), (float)w), scale3(ld3(T.v1), (float)u)), scale3(ld3(T.v2), (float)v));
        s.n = mk3(T.nx, T.ny, T.nz);
    } else {
        float4 sph = sc.spheres[h.prim - sc.n_tris];
        F3 center = mk3(sph.x, sph.y, sph.z);
        F3 phit = add3(o, scale3(d, (float)h.t));
        F3 dir = sub3(phit, center);
        dir = divs3(scale3(dir, sph.w), norm3(dir));
        F3 perr = scale3(abs3(dir), gamma_n(5));
        phit = add3(dir, center);
        perr = add3(scale3(perr, 1 + gamma_n(1)), scale3(add3(abs3(dir), abs3(center)), gamma_n(1)));
        s.p = phit;
        s.perr = perr;
        s.n = normalize3(sub3(phit, center));
    }
    s.wo = normalize3(neg3(d));
    int pm = sc.prim_mat[h.prim];
    s.mat = pm & 0xffff;
    s.light = (pm >> 16) - 1;
    return s;
}

// ------------------------------------------------------------------ lights
TKD int sample_discrete(const float* cdf, int n, float u, float* pdf) {   // src/TkMath.h:298-310, src/core/sampling.h:67-72
    int sz = n + 1;
    int size = sz - 2, first = 1;
    while (size > 0) {
        int half = size >> 1, middle = first + half;
        bool pr = cdf[middle] <= u;
        first = pr ? middle + 1 : first;
        size = pr ? size - (half + 1) : half;
    }
    int off = first - 1;
    off = off < 0 ? 0 : (off > sz - 2 ? sz - 2 : off);
    *pdf = cdf[off + 1] - cdf[off];
    return off;
}
struct LightPoint {
    F3 p, n, perr;
};
// Triangle::Sample (src/Triangle.cpp:56-66) on triangle `tri` of the scene.
TKD LightPoint tri_sample(const DScene& sc, int tri, float ux, float uy, float* pdf) {
    const TriShade T = sc.tri_shade[tri];
    F3 v0 = ld3(T.v0), v1 = ld3(T.v1), v2 = ld3(T.v2);
    float su0 = sqrtf(ux);
    float bx = 1 - su0, by = uy * su0;
    LightPoint r;
    r.p = add3(add3(scale3(v0, bx), scale3(v1, by)), scale3(v2, 1 - bx - by));
    r.n = mk3(T.nx, T.ny, T.nz);
    F3 s = add3(add3(abs3(scale3(v0, bx)), abs3(scale3(v1, by))), abs3(scale3(v2, 1 - bx - by)));
    r.perr = scale3(s, gamma_n(6));
    // Triangle ctor: area = |e1 x e2| * 0.5 (src/Triangle.hpp:21-24)
    float area = norm3(cross3(sub3(v1, v0), sub3(v2, v0))) * 0.5f;
    *pdf = 1 / area;
    return r;
}
TKD F3 light_L(const DLight& l, F3 n, F3 w) {                  // src/Light.cpp:90-93
    return (l.two_sided || dot3(n, w) > 0) ? ld3(l.Le) : mk3(0, 0, 0);
}

// Scene::uniformSampleOneLight (src/core/Scene.cpp:43-61) up to its shadow ray.  The light is picked by
// u_choice, the triangle of its mesh by u_tri and the point on it by (ulx, uly) (AreaLight::sample_Li,
// src/Light.cpp:24-39); the shadow ray is Intersection::spawnRayTo(other) (src/Intersection.hpp:40-48);
// `ld` is what the sample adds when that ray reaches the light.  f() of mirror and glass is black: a
// surface that is not diffuse gets no sample (and needs no shadow ray).
struct LightSample {
    bool want_shadow;
    F3 o, d, ld;
};
TKD LightSample sample_one_light(const DScene& sc, const SurfacePoint& sp, const TkMaterial& m, float u_choice, float ulx,
                                 float uly, float u_tri) {
    LightSample r;
    r.want_shadow = false;
    r.o = r.d = r.ld = mk3(0, 0, 0);
    const int nl = sc.n_lights;
    if (nl <= 0 || m.type != TK_DIFFUSE) return r;
    int li = (int)(u_choice * nl);
    li = li > nl - 1 ? nl - 1 : li;
    const float pdf_choice = 1.0f / nl;
    const DLight L = sc.lights[li];
    float pd, tmp;
    const int idx = sample_discrete(sc.light_cdf + L.cdf_first, L.tri_count, u_tri, &pd);
    const LightPoint lp = tri_sample(sc, L.tri_first + idx, ulx, uly, &tmp);
    const F3 wv = sub3(lp.p, sp.p);
    const float dist2 = dot3(wv, wv);
    if (dist2 == 0) tmp = 0;
    else { const F3 wn = normalize3(wv); tmp *= dist2 / absdot3(lp.n, neg3(wn)); }
    const float pdf_light = pd * tmp;
    if (pdf_light == 0 || dist2 == 0) return r;
    const F3 ws = normalize3(wv);
    const F3 Li = light_L(L, lp.n, neg3(ws));
    if (isblack(Li)) return r;
    r.o = offset_ray_origin(sp.p, sp.perr, sp.n, sub3(lp.p, sp.p));
    const F3 target = offset_ray_origin(lp.p, lp.perr, lp.n, sub3(r.o, lp.p));
    r.d = sub3(target, r.o);
    const F3 f = mat_f(m, sp.wo, ws, sp.n);
    r.ld = divs3(scale3(mul3(Li, f), absdot3(ws, sp.n)), pdf_light * pdf_choice);
    r.want_shadow = true;
    return r;
}

// AreaLight::sample_Le (src/Light.cpp:41-75) with MeshTriangle::Sample's
// triangle pick drawn from `upick` (src/Triangle.cpp:190-197): a point on the
// light, a cosine-distributed direction about its normal, the spawned ray.
struct LightEmission {
    F3 o, d, n, Le;
    float pdf_pos, pdf_dir;
};
TKD LightEmission light_sample_le(const DScene& sc, const DLight& L, float u1x, float u1y, float u2x, float u2y,
                                  float upick) {
    LightEmission e;
    float pd, tmp;
    const int idx = sample_discrete(sc.light_cdf + L.cdf_first, L.tri_count, upick, &pd);
    LightPoint lp = tri_sample(sc, L.tri_first + idx, u1x, u1y, &tmp);
    e.pdf_pos = pd * tmp;
    F3 w;
    if (L.two_sided) {
        float ux = u2x;
        if (ux < 0.5f) { ux = sse_min(ux * 2, kOneMinusEps); w = cosine_hemisphere(ux, u2y); }
        else { ux = sse_min((ux - 0.5f) * 2, kOneMinusEps); w = cosine_hemisphere(ux, u2y); w.z *= -1; }
        e.pdf_dir = 0.5f * (fabsf(w.z) * kInvPi);
    } else {
        w = cosine_hemisphere(u2x, u2y);
        e.pdf_dir = w.z * kInvPi;
    }
    F3 v1, v2;
    coordinate_system(lp.n, &v1, &v2);
    w = add3(add3(scale3(v1, w.x), scale3(v2, w.y)), scale3(lp.n, w.z));
    e.Le = light_L(L, lp.n, w);
    e.o = offset_ray_origin(lp.p, lp.perr, lp.n, w);
    e.d = w;
    e.n = lp.n;
    return e;
}

}  // namespace tkd
