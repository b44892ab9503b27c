/*
 * sppm_oracle.c — CPU restatement of the reference SPPM iteration, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see sppm_oracle.h).  Pinned to outputs of the
 * reference built headless (oracle/build_ref.sh, oracle/ref_probe.cpp,
 * tests/golden/ref_*.npz, tests/test_ref_pinned_cpu.py).  Every function
 * cites the reference lines it restates (paths relative to the upstream
 * repository root).  Build with -ffp-contract=off: the reference binary is
 * baseline x86-64 (no FMA), so products and sums round separately.
 *
 * Deliberate departures from the reference, all forced by its behaviour:
 *  - get_random_float() (src/TkMath.h:312-319) seeds a fresh mt19937 from
 *    std::random_device per call, so the reference has no reproducible random
 *    stream.  Here every such call is a counter-based hash keyed on
 *    (seed, path or pixel, bounce or iteration, slot); see rng_u01().
 *  - Photon paths are numbered like the worker threads number them
 *    (haltonIdx = ++shootGlobalPhotons, src/tracer/TkSppmTracer.cpp:414) and
 *    stored in path order; the reference interleaves threads under a mutex.
 *    Caustic path j uses index j + (global paths shot in earlier iterations);
 *    the reference adds the global count at the time of the call (:419), which
 *    includes the current iteration and would chain the two passes.
 *  - Every tile of the image is traced (the reference skips the trailing
 *    tiles when width or height is not a multiple of 20,
 *    src/core/TkRayTracer.cpp:115-117).
 */
#define _GNU_SOURCE
#include "sppm_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ math */
typedef struct { float x, y, z; } v3;

static float k_pi, k_inv_pi, k_two_pi, k_half_pi;
static const float k_eps = FLT_EPSILON * 0.5f;           /* src/TkMath.cpp:14 */
static const float k_one_minus_eps = 0x1.fffffep-1f;     /* src/TkMath.cpp:15 */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void init_consts(void) {
    k_pi = (float)(4.0 * atan(1.0));          /* src/TkMath.cpp:7  */
    k_inv_pi = 1.0f / k_pi;                   /* src/TkMath.cpp:8  */
    k_two_pi = (float)(2.0 * k_pi);           /* src/TkMath.cpp:9  */
    k_half_pi = (float)(0.5 * k_pi);          /* src/TkMath.cpp:10 */
}
static void ensure_consts(void) { pthread_once(&g_once, init_consts); }

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vabs(v3 a) { return V(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
/* src/Vector.hpp:91-94 */
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* src/Vector.hpp:96-103 */
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* src/Vector.hpp:80-89 */
static inline v3 vnormalize(v3 v) {
    float mag2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (mag2 > 0) {
        float inv = 1 / sqrtf(mag2);
        return V(v.x * inv, v.y * inv, v.z * inv);
    }
    return v;
}
static inline float vnorm(v3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
static inline float vabsdot(v3 a, v3 b) { return fabsf(a.x * b.x + a.y * b.y + a.z * b.z); }
static inline float vget(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* src/TkMath.h:11-23: SSE min/max return the second operand on NaN. */
static inline float tk_min(float a, float b) { return a < b ? a : b; }
static inline float tk_max(float a, float b) { return a > b ? a : b; }
/* src/TkMath.h:201-204 */
static inline float gamma_n(int n) { return (n * k_eps) / (1 - n * k_eps); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
/* src/TkMath.h:334-346 */
static inline float next_up(float v) {
    if (isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = f2u(v);
    if (v >= 0) ++ui; else --ui;
    return u2f(ui);
}
/* src/TkMath.h:348-358 */
static inline float next_down(float v) {
    if (isinf(v) && v < 0.) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = f2u(v);
    if (v > 0) --ui; else ++ui;
    return u2f(ui);
}

/* src/Vector.hpp:116-123 */
static inline void coordinate_system(v3 v1, v3* v2, v3* v3o) {
    float sign = copysignf(1.0f, v1.z);
    float a = -1 / (sign + v1.z);
    float b = v1.x * v1.y * a;
    *v2 = V(1 + sign * v1.x * v1.x * a, sign * b, -sign * v1.x);
    *v3o = V(b, sign + v1.y * v1.y * a, -v1.y);
}

/* ------------------------------------------------------------------- rng */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
/* Counter-based replacement for get_random_float(): 24 uniform bits in [0,1).
 * The GPU kernels implement the same function (written independently there). */
float orc_rng_u01(uint32_t seed, uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = seed ^ 0x9E3779B9u;
    h = mix32(h + a * 0x85EBCA6Bu);
    h = mix32(h + b * 0xC2B2AE35u);
    h = mix32(h + c * 0x27D4EB2Fu);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

/* Cost model of get_random_float (src/TkMath.h:312-319): constructing
 * std::mt19937 runs the 624-word seeding recurrence and the first draw runs
 * one full twist. */
static uint32_t mt_cost(uint32_t seed) {
    uint32_t mt[624];
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    for (int i = 0; i < 624; ++i) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    uint32_t y = mt[0];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
static volatile uint32_t g_mt_sink;

/* ----------------------------------------------------- low discrepancy */
static const int32_t k_primes[16] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53};

/* src/core/TkLowdiscrepancy.h:19-32 */
float orc_radical_inverse(int32_t base_index, uint64_t a) {
    int32_t base = k_primes[base_index & 15];
    float inv_base = (float)1 / (float)base, inv_base_n = 1;
    uint64_t reversed = 0;
    while (a) {
        uint64_t next = a / (uint64_t)base;
        uint64_t digit = a - next * (uint64_t)base;
        reversed = reversed * (uint64_t)base + digit;
        inv_base_n *= inv_base;
        a = next;
    }
    return tk_min((float)reversed * inv_base_n, k_one_minus_eps);
}

/* src/TkHash.h:10-63 (MurmurHash64A) applied to the 16 bytes
 * {pixel.x, pixel.y, dim, seed} as Hash(currentPixel, dim, seed) packs them
 * (src/TkHash.h:111-119, src/core/sampler.cpp:48). */
uint64_t orc_hash_pixel(int32_t x, int32_t y, int32_t dim, int32_t seed) {
    const uint64_t m = 0xc6a4a7935bd1e995ull;
    const int r = 47;
    uint64_t h = 0 ^ (16 * m);
    uint64_t blk[2];
    blk[0] = (uint64_t)(uint32_t)x | ((uint64_t)(uint32_t)y << 32);
    blk[1] = (uint64_t)(uint32_t)dim | ((uint64_t)(uint32_t)seed << 32);
    for (int i = 0; i < 2; ++i) {
        uint64_t k = blk[i];
        k *= m; k ^= k >> r; k *= m;
        h ^= k; h *= m;
    }
    h ^= h >> r; h *= m; h ^= h >> r;
    return h;
}

/* src/TkMath.h:206-234 */
int32_t orc_permutation_element(uint32_t i, uint32_t l, uint32_t p) {
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893d; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8;
        i *= 0x0929eb3f; i ^= p >> 23; i ^= (i & w) >> 1; i *= 1 | p >> 27;
        i *= 0x6935fa69; i ^= (i & w) >> 11; i *= 0x74dcb303; i ^= (i & w) >> 2;
        i *= 0x9e501cc3; i ^= (i & w) >> 2; i *= 0xc860a3df; i &= w; i ^= i >> 5;
    } while (i >= l);
    return (int32_t)((i + p) % l);
}

/* ------------------------------------------------------------- sampling */
/* Correctly rounded float sine / cosine / arc functions: the double function,
 * rounded to float.  The reference's unqualified cos / sin / acos / atan2 on
 * float arguments resolve to the float overloads (MSVC, and g++ once <math.h> is
 * included); a float libm's last bit differs from platform to platform, this
 * form gives the same value on the host and on the GPU. */
static inline float tk_cosf(float x) { return (float)cos((double)x); }
static inline float tk_sinf(float x) { return (float)sin((double)x); }
static inline float tk_acosf(float x) { return (float)acos((double)x); }
static inline float tk_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
/* src/core/sampling.cpp:60-77: float cos / sin, float products. */
static void concentric_disk(float ux, float uy, float* dx, float* dy) {
    float ox = ux * 2.f - 1.0f, oy = uy * 2.f - 1.0f;
    if (ox == 0.0f && oy == 0.f) { *dx = ox; *dy = oy; return; }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) {
        r = ox;
        theta = (float)(k_half_pi * 0.5 * (oy / ox));
    } else {
        r = oy;
        theta = (float)(k_half_pi - k_half_pi * 0.5 * (ox / oy));
    }
    *dx = r * tk_cosf(theta);
    *dy = r * tk_sinf(theta);
}
/* src/core/sampling.h:32-37 */
static v3 cosine_hemisphere(float ux, float uy) {
    float dx, dy;
    concentric_disk(ux, uy, &dx, &dy);
    float z = sqrtf(tk_max(0.f, 1 - dx * dx - dy * dy));
    return V(dx, dy, z);
}

/* --------------------------------------------------------------- scene */
typedef struct { v3 mn, mx; } box3;
static inline box3 box_empty(void) {
    box3 b; b.mn = V(INFINITY, INFINITY, INFINITY); b.mx = V(-INFINITY, -INFINITY, -INFINITY);
    return b;                                            /* src/Bounds3.hpp:17-23 */
}
static inline box3 box_union(box3 a, box3 b) {           /* src/Bounds3.hpp:127-133 */
    box3 r;
    r.mn = V(fminf(a.mn.x, b.mn.x), fminf(a.mn.y, b.mn.y), fminf(a.mn.z, b.mn.z));
    r.mx = V(fmaxf(a.mx.x, b.mx.x), fmaxf(a.mx.y, b.mx.y), fmaxf(a.mx.z, b.mx.z));
    return r;
}
static inline box3 box_union_p(box3 a, v3 p) {
    box3 b; b.mn = p; b.mx = p; return box_union(a, b);
}
static inline int box_max_extent(box3 b) {               /* src/Bounds3.hpp:32-41 */
    v3 d = vsub(b.mx, b.mn);
    if (d.x > d.y && d.x > d.z) return 0;
    else if (d.y > d.z) return 1;
    return 2;
}
static inline double box_area(box3 b) {                  /* src/Bounds3.hpp:43-47 */
    v3 d = vsub(b.mx, b.mn);
    return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
}
static inline v3 box_centroid(box3 b) {                  /* src/Bounds3.hpp:49 */
    return vadd(vscale(b.mn, 0.5f), vscale(b.mx, 0.5f));
}

typedef struct {
    v3 v0, v1, v2, e1, e2, normal;
    float area;
} Tri;                                                   /* src/Triangle.hpp:12-25 */

typedef struct {
    box3 bounds;
    int32_t offset;    /* primitivesOffset or secondChildOffset */
    uint16_t nprims;
    uint8_t axis;
} LNode;                                                 /* src/BVH.cpp:49-60 */

typedef struct {
    LNode* nodes;
    int32_t n_nodes;
    int32_t* prims;    /* ordered primitive indices */
    int32_t n_prims;
} Bvh;

typedef struct { v3 o, d, dinv; double tmax; } Ray;      /* src/Ray.hpp:11-24 */

typedef struct {
    v3 p, perr, wo, n;
    int32_t obj;
    int32_t prim;
} Isect;                                                 /* src/Intersection.hpp:12-19 */

typedef struct {
    float* cdf;
    int n;
} Dist1D;                                                /* src/core/sampling.h:53-73 */

struct orc_scene {
    TkObject* objects; int n_objects;
    Tri* tris; int n_tris;
    TkSphere* spheres; int n_spheres;
    TkMaterial* mats; int n_mats;
    TkLight* lights; int n_lights;
    Bvh top;           /* over objects: src/core/Scene.cpp:18-20 */
    Bvh* mesh_bvh;     /* per object (mesh only): src/Triangle.cpp:179 */
    Dist1D* light_dist;
    float* light_area;
    float c2w[16], r2c[16];
    int w, h;
};

static inline Ray make_ray(v3 o, v3 d) {                 /* src/Ray.hpp:20-24 */
    Ray r; r.o = o; r.d = d;
    r.dinv = V((float)(1. / d.x), (float)(1. / d.y), (float)(1. / d.z));
    r.tmax = DBL_MAX;
    return r;
}

/* src/Ray.hpp:43-57 */
static v3 offset_ray_origin(v3 p, v3 perr, v3 n, v3 w) {
    float d = vdot(vabs(n), perr);
    v3 off = vscale(n, d);
    if (vdot(w, n) < 0) off = vneg(off);
    v3 po = vadd(p, off);
    float* pf = &po.x; float* of = &off.x;
    for (int i = 0; i < 3; ++i) {
        if (of[i] > 0) pf[i] = next_up(pf[i]);
        else if (of[i] < 0) pf[i] = next_down(pf[i]);
    }
    return po;
}
/* src/Intersection.hpp:25-29 */
static Ray spawn_ray(const Isect* is, v3 dir) {
    return make_ray(offset_ray_origin(is->p, is->perr, is->n, dir), dir);
}
/* src/Intersection.hpp:40-48 */
static Ray spawn_ray_to(const Isect* is, const Isect* other) {
    v3 o = offset_ray_origin(is->p, is->perr, is->n, vsub(other->p, is->p));
    v3 target = offset_ray_origin(other->p, other->perr, other->n, vsub(o, other->p));
    Ray r = make_ray(o, vsub(target, o));
    r.tmax = k_one_minus_eps;
    return r;
}

/* src/Triangle.cpp:15-52 */
static int tri_intersect(const Tri* T, Ray* r, Isect* is) {
    double u, v, t_tmp = 0;
    v3 pvec = vcross(r->d, T->e2);
    double det = vdot(T->e1, pvec);
    if (fabs(det) < k_eps) return 0;
    double det_inv = 1. / det;
    v3 tvec = vsub(r->o, T->v0);
    u = vdot(tvec, pvec) * det_inv;
    if (u < 0 || u > 1) return 0;
    v3 qvec = vcross(tvec, T->e1);
    v = vdot(r->d, qvec) * det_inv;
    if (v < 0 || u + v > 1) return 0;
    t_tmp = vdot(T->e2, qvec) * det_inv;
    double w = 1 - u - v;
    if (t_tmp > 0 && t_tmp < r->tmax) {
        float xs = (float)(fabs(w * T->v0.x) + fabs(u * T->v1.x) + fabs(v * T->v2.x));
        float ys = (float)(fabs(w * T->v0.y) + fabs(u * T->v1.y) + fabs(v * T->v2.y));
        float zs = (float)(fabs(w * T->v0.z) + fabs(u * T->v1.z) + fabs(v * T->v2.z));
        is->perr = vscale(V(xs, ys, zs), gamma_n(7));
        is->p = vadd(vadd(vscale(T->v0, (float)w), vscale(T->v1, (float)u)), vscale(T->v2, (float)v));
        is->wo = vnormalize(vneg(r->d));
        is->n = T->normal;
        r->tmax = t_tmp;
        return 1;
    }
    return 0;
}

/* src/EFloat.h:8-144 */
typedef struct { float v, lo, hi; } ef;
static inline ef ef_make(float v, float err) {
    ef r; r.v = v;
    if (err == 0.0f) r.lo = r.hi = v;
    else { r.lo = next_down(v - err); r.hi = next_up(v + err); }
    return r;
}
static inline ef ef_add(ef a, ef b) {
    ef r; r.v = a.v + b.v; r.lo = next_down(a.lo + b.lo); r.hi = next_up(a.hi + b.hi); return r;
}
static inline ef ef_sub(ef a, ef b) {
    ef r; r.v = a.v - b.v; r.lo = next_down(a.lo - b.hi); r.hi = next_up(a.hi - b.lo); return r;
}
static inline float min4(float a, float b, float c, float d) {
    float m0 = a < b ? a : b, m1 = c < d ? c : d;   /* std::min semantics */
    (void)0;
    return m1 < m0 ? m1 : m0;
}
static inline float max4(float a, float b, float c, float d) {
    float m0 = a < b ? b : a, m1 = c < d ? d : c;   /* std::max semantics */
    return m0 < m1 ? m1 : m0;
}
static inline ef ef_mul(ef a, ef b) {
    ef r; r.v = a.v * b.v;
    float p0 = a.lo * b.lo, p1 = a.hi * b.lo, p2 = a.lo * b.hi, p3 = a.hi * b.hi;
    r.lo = next_down(min4(p0, p1, p2, p3));
    r.hi = next_up(max4(p0, p1, p2, p3));
    return r;
}
static inline ef ef_div(ef a, ef b) {
    ef r; r.v = a.v / b.v;
    if (b.lo < 0 && b.hi > 0) { r.lo = -INFINITY; r.hi = INFINITY; }
    else {
        float d0 = a.lo / b.lo, d1 = a.hi / b.lo, d2 = a.lo / b.hi, d3 = a.hi / b.hi;
        r.lo = next_down(min4(d0, d1, d2, d3));
        r.hi = next_up(max4(d0, d1, d2, d3));
    }
    return r;
}
/* src/EFloat.h:125-143 */
static int ef_quadratic(ef a, ef b, ef c, ef* t0, ef* t1) {
    float discrim = b.v * b.v - 4.0f * a.v * c.v;
    if (discrim < 0) return 0;
    float root = sqrtf(discrim);
    ef froot = ef_make(root, k_eps * root);
    ef q;
    if (b.v < 0) q = ef_mul(ef_sub(b, froot), ef_make(-0.5f, 0));
    else q = ef_mul(ef_add(b, froot), ef_make(-0.5f, 0));
    *t0 = ef_div(q, a);
    *t1 = ef_div(c, q);
    if (t0->v > t1->v) { ef t = *t0; *t0 = *t1; *t1 = t; }
    return 1;
}

/* src/Sphere.cpp:14-55 */
static int sphere_intersect(const TkSphere* S, Ray* r, Isect* is) {
    v3 center = ld3(S->center);
    ef ox = ef_make(r->o.x, 0), oy = ef_make(r->o.y, 0), oz = ef_make(r->o.z, 0);
    ef cx = ef_make(center.x, 0), cy = ef_make(center.y, 0), cz = ef_make(center.z, 0);
    ox = ef_sub(ox, cx); oy = ef_sub(oy, cy); oz = ef_sub(oz, cz);
    ef dx = ef_make(r->d.x, 0), dy = ef_make(r->d.y, 0), dz = ef_make(r->d.z, 0);
    ef a = ef_add(ef_add(ef_mul(dx, dx), ef_mul(dy, dy)), ef_mul(dz, dz));
    ef b = ef_mul(ef_add(ef_add(ef_mul(dx, ox), ef_mul(dy, oy)), ef_mul(dz, oz)), ef_make(2.0f, 0));
    ef rr = ef_make(S->radius, 0);
    ef c = ef_sub(ef_add(ef_add(ef_mul(ox, ox), ef_mul(oy, oy)), ef_mul(oz, oz)), ef_mul(rr, rr));
    ef t0, t1;
    if (!ef_quadratic(a, b, c, &t0, &t1)) return 0;
    if (t0.hi > r->tmax || t1.lo <= 0) return 0;
    ef hit = t0;
    if (hit.lo <= 0) {
        hit = t1;
        if (hit.hi > r->tmax) return 0;
    }
    v3 phit = vadd(r->o, vscale(r->d, hit.v));
    v3 dir = vsub(phit, center);
    dir = vdivs(vscale(dir, S->radius), vnorm(dir));
    v3 perr = vscale(vabs(dir), gamma_n(5));
    phit = vadd(dir, center);
    perr = vadd(vscale(perr, 1 + gamma_n(1)), vscale(vadd(vabs(dir), vabs(center)), gamma_n(1)));
    is->p = phit;
    is->perr = perr;
    is->wo = vnormalize(vneg(r->d));
    is->n = vnormalize(vsub(phit, center));
    r->tmax = (double)hit.v;
    return 1;
}

/* src/Bounds3.hpp:96-125.  dir_pos[i] = direction[i] > 0 (src/BVH.cpp:249). */
static inline int box_intersect_p(const box3* b, const Ray* r, const int dir_pos[3]) {
    v3 tmp[2];
    tmp[0] = vmul(vsub(b->mx, r->o), r->dinv);
    tmp[1] = vmul(vsub(b->mn, r->o), r->dinv);
    float tmin = tmp[dir_pos[0]].x, tmaxv = tmp[1 - dir_pos[0]].x;
    float tymin = tmp[dir_pos[1]].y, tymax = tmp[1 - dir_pos[1]].y;
    const float pad = 1 + 2 * gamma_n(3);
    tmaxv *= pad; tymax *= pad;
    if (tmin > tymax || tymin > tmaxv) return 0;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmaxv) tmaxv = tymax;
    float tzmin = tmp[dir_pos[2]].z, tzmax = tmp[1 - dir_pos[2]].z;
    tzmax *= pad;
    if (tmin > tzmax || tzmin > tmaxv) return 0;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmaxv) tmaxv = tzmax;
    return (tmin <= tmaxv) && (tmaxv > 0);
}
/* The same slab test that also rejects boxes entirely beyond the current hit. */
static inline int box_intersect_cull(const box3* b, const Ray* r, const int dir_pos[3], float* tnear) {
    v3 tmp[2];
    tmp[0] = vmul(vsub(b->mx, r->o), r->dinv);
    tmp[1] = vmul(vsub(b->mn, r->o), r->dinv);
    float tmin = tmp[dir_pos[0]].x, tmaxv = tmp[1 - dir_pos[0]].x;
    float tymin = tmp[dir_pos[1]].y, tymax = tmp[1 - dir_pos[1]].y;
    const float pad = 1 + 2 * gamma_n(3);
    tmaxv *= pad; tymax *= pad;
    if (tmin > tymax || tymin > tmaxv) return 0;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmaxv) tmaxv = tymax;
    float tzmin = tmp[dir_pos[2]].z, tzmax = tmp[1 - dir_pos[2]].z;
    tzmax *= pad;
    if (tmin > tzmax || tzmin > tmaxv) return 0;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmaxv) tmaxv = tzmax;
    *tnear = tmin;
    return (tmin <= tmaxv) && (tmaxv > 0) && ((double)tmin < r->tmax);
}

/* ------------------------------------------------------------ BVH build */
typedef struct { int32_t index; box3 bounds; v3 centroid; } PrimInfo;   /* src/BVH.cpp:32-41 */
typedef struct BNode {
    box3 bounds; struct BNode *left, *right; int axis, first, nprims;
} BNode;                                                                 /* src/BVH.cpp:8-30 */

typedef struct {
    PrimInfo* info;
    int32_t* ordered; int n_ordered;
    BNode* pool; int pool_used, pool_cap;
    int max_prims;
} BuildCtx;

static BNode* bnode_new(BuildCtx* c) {
    if (c->pool_used == c->pool_cap) { fprintf(stderr, "oracle: bvh pool exhausted\n"); abort(); }
    BNode* n = &c->pool[c->pool_used++];
    memset(n, 0, sizeof(*n));
    return n;
}
static void bnode_leaf(BuildCtx* c, BNode* node, int start, int end, box3 bounds) {
    node->first = c->n_ordered;
    for (int i = start; i < end; ++i) c->ordered[c->n_ordered++] = c->info[i].index;
    node->nprims = end - start;
    node->bounds = bounds;
    node->left = node->right = 0;
}

/* src/BVH.cpp:90-223 (SplitMethod::SAH) */
static BNode* bvh_recursive(BuildCtx* c, int start, int end) {
    BNode* node = bnode_new(c);
    PrimInfo* pi = c->info;
    box3 bounds = box_empty();
    for (int i = start; i < end; ++i) bounds = box_union(bounds, pi[i].bounds);
    int nprims = end - start;
    if (nprims == 1) { bnode_leaf(c, node, start, end, bounds); return node; }
    box3 cb = box_empty();
    for (int i = start; i < end; ++i) cb = box_union_p(cb, pi[i].centroid);
    int mid = (start + end) / 2;
    int dim = box_max_extent(cb);
    v3 diag = vsub(cb.mx, cb.mn);
    if (vget(cb.mn, dim) == vget(cb.mx, dim)) { bnode_leaf(c, node, start, end, bounds); return node; }
    if (nprims <= 2) {
        /* nth_element on two elements: order them by centroid[dim] */
        if (vget(pi[start + 1].centroid, dim) < vget(pi[start].centroid, dim)) {
            PrimInfo t = pi[start]; pi[start] = pi[start + 1]; pi[start + 1] = t;
        }
    } else {
        enum { NB = 12 };
        int count[NB]; box3 bb[NB];
        for (int i = 0; i < NB; ++i) { count[i] = 0; bb[i] = box_empty(); }
        float cmin = vget(cb.mn, dim), dd = vget(diag, dim);
        for (int i = start; i < end; ++i) {
            int b = (int)(NB * ((vget(pi[i].centroid, dim) - cmin) / dd));
            if (b == NB) b = NB - 1;
            count[b]++;
            bb[b] = box_union_p(bb[b], pi[i].centroid);
        }
        double cost[NB - 1];
        for (int i = 0; i < NB - 1; ++i) {
            box3 b0 = box_empty(), b1 = box_empty();
            int c0 = 0, c1 = 0;
            for (int j = 0; j <= i; ++j) { b0 = box_union(b0, bb[j]); c0 += count[j]; }
            for (int j = i + 1; j < NB; ++j) { b1 = box_union(b1, bb[j]); c1 += count[j]; }
            cost[i] = 1 + (c0 * box_area(b0) + c1 * box_area(b1)) / box_area(bounds);
        }
        double min_cost = cost[0]; int min_b = 0;
        for (int i = 0; i < NB - 1; ++i) if (cost[i] < min_cost) { min_cost = cost[i]; min_b = i; }
        double leaf_cost = nprims;
        if (nprims > c->max_prims || min_cost < leaf_cost) {
            int l = start, r = end - 1;      /* std::partition: predicate-true first */
            while (1) {
                while (l <= r) {
                    int b = (int)(NB * ((vget(pi[l].centroid, dim) - cmin) / dd));
                    if (b == NB) b = NB - 1;
                    if (b <= min_b) ++l; else break;
                }
                while (l < r) {
                    int b = (int)(NB * ((vget(pi[r].centroid, dim) - cmin) / dd));
                    if (b == NB) b = NB - 1;
                    if (!(b <= min_b)) --r; else break;
                }
                if (l >= r) break;
                PrimInfo t = pi[l]; pi[l] = pi[r]; pi[r] = t;
                ++l; --r;
            }
            mid = l;
        } else { bnode_leaf(c, node, start, end, bounds); return node; }
    }
    BNode* L = bvh_recursive(c, start, mid);
    BNode* R = bvh_recursive(c, mid, end);
    node->left = L; node->right = R;
    node->bounds = box_union(L->bounds, R->bounds);
    node->axis = dim; node->nprims = 0;
    return node;
}
/* src/BVH.cpp:225-243 */
static int bvh_flatten(Bvh* out, BNode* node, int* offset) {
    LNode* ln = &out->nodes[*offset];
    ln->bounds = node->bounds;
    int my = (*offset)++;
    if (node->nprims > 0) { ln->offset = node->first; ln->nprims = (uint16_t)node->nprims; ln->axis = 0; }
    else {
        ln->axis = (uint8_t)node->axis; ln->nprims = 0;
        bvh_flatten(out, node->left, offset);
        ln->offset = bvh_flatten(out, node->right, offset);
    }
    return my;
}
/* src/BVH.cpp:62-83 */
static void bvh_build(Bvh* out, const box3* prim_bounds, int n, int max_prims) {
    memset(out, 0, sizeof(*out));
    if (n == 0) return;
    BuildCtx c;
    c.info = (PrimInfo*)malloc(sizeof(PrimInfo) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        c.info[i].index = i; c.info[i].bounds = prim_bounds[i];
        c.info[i].centroid = box_centroid(prim_bounds[i]);
    }
    c.ordered = (int32_t*)malloc(sizeof(int32_t) * (size_t)n); c.n_ordered = 0;
    c.pool_cap = 2 * n + 1; c.pool_used = 0;
    c.pool = (BNode*)malloc(sizeof(BNode) * (size_t)c.pool_cap);
    c.max_prims = max_prims < 255 ? max_prims : 255;
    BNode* root = bvh_recursive(&c, 0, n);
    out->nodes = (LNode*)malloc(sizeof(LNode) * (size_t)c.pool_used);
    out->n_nodes = c.pool_used;
    int off = 0;
    bvh_flatten(out, root, &off);
    out->prims = c.ordered; out->n_prims = n;
    free(c.info); free(c.pool);
}

/* ---------------------------------------------------- scene construction */
static box3 tri_bounds(const Tri* t) {                   /* src/Triangle.cpp:54 */
    box3 b;
    b.mn = V(fminf(t->v0.x, t->v1.x), fminf(t->v0.y, t->v1.y), fminf(t->v0.z, t->v1.z));
    b.mx = V(fmaxf(t->v0.x, t->v1.x), fmaxf(t->v0.y, t->v1.y), fmaxf(t->v0.z, t->v1.z));
    return box_union_p(b, t->v2);
}

orc_scene* orc_scene_create(const TkObject* objects, int n_objects, const TkTriangle* tris,
                            int n_tris, const TkSphere* spheres, int n_spheres,
                            const TkMaterial* materials, int n_materials,
                            const TkLight* lights, int n_lights) {
    ensure_consts();
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    s->n_objects = n_objects; s->n_tris = n_tris; s->n_spheres = n_spheres;
    s->n_mats = n_materials; s->n_lights = n_lights;
    s->objects = (TkObject*)malloc(sizeof(TkObject) * (size_t)(n_objects ? n_objects : 1));
    memcpy(s->objects, objects, sizeof(TkObject) * (size_t)n_objects);
    s->spheres = (TkSphere*)malloc(sizeof(TkSphere) * (size_t)(n_spheres ? n_spheres : 1));
    if (n_spheres) memcpy(s->spheres, spheres, sizeof(TkSphere) * (size_t)n_spheres);
    s->mats = (TkMaterial*)malloc(sizeof(TkMaterial) * (size_t)(n_materials ? n_materials : 1));
    memcpy(s->mats, materials, sizeof(TkMaterial) * (size_t)n_materials);
    s->lights = (TkLight*)malloc(sizeof(TkLight) * (size_t)(n_lights ? n_lights : 1));
    if (n_lights) memcpy(s->lights, lights, sizeof(TkLight) * (size_t)n_lights);
    s->tris = (Tri*)malloc(sizeof(Tri) * (size_t)(n_tris ? n_tris : 1));
    for (int i = 0; i < n_tris; ++i) {                   /* src/Triangle.hpp:18-25 */
        Tri* t = &s->tris[i];
        t->v0 = ld3(tris[i].v0); t->v1 = ld3(tris[i].v1); t->v2 = ld3(tris[i].v2);
        t->e1 = vsub(t->v1, t->v0); t->e2 = vsub(t->v2, t->v0);
        t->normal = vnormalize(vcross(t->e1, t->e2));
        t->area = vnorm(vcross(t->e1, t->e2)) * 0.5f;
    }
    /* per-mesh BVH over its triangles (src/Triangle.cpp:169-179) and the
     * object bounds for the top-level BVH (src/Triangle.cpp:106-167) */
    s->mesh_bvh = (Bvh*)calloc((size_t)(n_objects ? n_objects : 1), sizeof(Bvh));
    box3* obounds = (box3*)malloc(sizeof(box3) * (size_t)(n_objects ? n_objects : 1));
    for (int o = 0; o < n_objects; ++o) {
        const TkObject* ob = &s->objects[o];
        if (ob->kind == 0) {
            box3* tb = (box3*)malloc(sizeof(box3) * (size_t)(ob->count ? ob->count : 1));
            box3 all = box_empty();
            for (int i = 0; i < ob->count; ++i) {
                const Tri* t = &s->tris[ob->first + i];
                tb[i] = tri_bounds(t);
                all = box_union_p(box_union_p(box_union_p(all, t->v0), t->v1), t->v2);
            }
            bvh_build(&s->mesh_bvh[o], tb, ob->count, 1);
            free(tb);
            obounds[o] = all;
        } else {                                         /* src/Sphere.cpp:76-80 */
            const TkSphere* sp = &s->spheres[ob->first];
            obounds[o].mn = V(sp->center[0] - sp->radius, sp->center[1] - sp->radius, sp->center[2] - sp->radius);
            obounds[o].mx = V(sp->center[0] + sp->radius, sp->center[1] + sp->radius, sp->center[2] + sp->radius);
        }
    }
    bvh_build(&s->top, obounds, n_objects, 1);
    free(obounds);
    /* Distribution1D over triangle areas per light (src/Triangle.cpp:170-178,
     * src/core/sampling.h:57-66) */
    s->light_dist = (Dist1D*)calloc((size_t)(n_lights ? n_lights : 1), sizeof(Dist1D));
    s->light_area = (float*)calloc((size_t)(n_lights ? n_lights : 1), sizeof(float));
    for (int l = 0; l < n_lights; ++l) {
        const TkObject* ob = &s->objects[s->lights[l].object];
        int num = ob->count;
        Dist1D* d = &s->light_dist[l];
        d->n = num;
        d->cdf = (float*)malloc(sizeof(float) * (size_t)(num + 1));
        d->cdf[0] = 0;
        float area = 0;
        for (int i = 1; i < num + 1; ++i) {
            float f = s->tris[ob->first + i - 1].area;
            area += f;
            d->cdf[i] = d->cdf[i - 1] + f / num;
        }
        float c = d->cdf[num];
        for (int i = 1; i < num + 1; ++i) d->cdf[i] /= c;
        s->light_area[l] = area;
    }
    return s;
}

void orc_scene_destroy(orc_scene* s) {
    if (!s) return;
    for (int o = 0; o < s->n_objects; ++o) { free(s->mesh_bvh[o].nodes); free(s->mesh_bvh[o].prims); }
    for (int l = 0; l < s->n_lights; ++l) free(s->light_dist[l].cdf);
    free(s->mesh_bvh); free(s->top.nodes); free(s->top.prims);
    free(s->light_dist); free(s->light_area);
    free(s->objects); free(s->tris); free(s->spheres); free(s->mats); free(s->lights);
    free(s);
}

void orc_scene_set_camera(orc_scene* s, const float c2w[16], const float raster2cam[16], int w, int h) {
    memcpy(s->c2w, c2w, sizeof(float) * 16);
    memcpy(s->r2c, raster2cam, sizeof(float) * 16);
    s->w = w; s->h = h;
}

/* ------------------------------------------------------------ traversal */
typedef struct { int64_t rays, nodes, prims; } TraceCounters;

/* src/BVH.cpp:245-288 for a mesh BVH whose leaves are triangles. */
static int mesh_intersect(const orc_scene* s, int obj, Ray* r, Isect* is, int faithful, TraceCounters* tc) {
    const Bvh* b = &s->mesh_bvh[obj];
    if (!b->nodes) return 0;
    const int first = s->objects[obj].first;
    int hit = 0;
    int dir_pos[3] = {r->d.x > 0, r->d.y > 0, r->d.z > 0};
    int to_visit = 0, cur = 0;
    int stack[64];
    while (1) {
        const LNode* node = &b->nodes[cur];
        float tn;
        int in = faithful ? box_intersect_p(&node->bounds, r, dir_pos)
                          : box_intersect_cull(&node->bounds, r, dir_pos, &tn);
        if (tc) tc->nodes++;
        if (in) {
            if (node->nprims > 0) {
                for (int i = 0; i < node->nprims; ++i) {
                    int ti = first + b->prims[node->offset + i];
                    if (tc) tc->prims++;
                    if (tri_intersect(&s->tris[ti], r, is)) { hit = 1; is->prim = ti; }
                }
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            } else {
                /* reference: far child first (src/BVH.cpp:269-278) */
                int far_first = faithful ? 1 : 0;
                int go_second = dir_pos[node->axis] ? far_first : !far_first;
                if (go_second) { stack[to_visit++] = cur + 1; cur = node->offset; }
                else { stack[to_visit++] = node->offset; cur = cur + 1; }
            }
        } else {
            if (to_visit == 0) break;
            cur = stack[--to_visit];
        }
    }
    return hit;
}

/* Scene::intersect (src/core/Scene.cpp:22-25): top-level BVH over objects,
 * Object::intersect (src/core/Object.cpp:11-16) per leaf. */
static int scene_intersect(const orc_scene* s, Ray* r, Isect* is, int faithful, TraceCounters* tc) {
    const Bvh* b = &s->top;
    if (tc) tc->rays++;
    if (!b->nodes) return 0;
    int hit = 0;
    int dir_pos[3] = {r->d.x > 0, r->d.y > 0, r->d.z > 0};
    int to_visit = 0, cur = 0;
    int stack[64];
    while (1) {
        const LNode* node = &b->nodes[cur];
        float tn;
        int in = faithful ? box_intersect_p(&node->bounds, r, dir_pos)
                          : box_intersect_cull(&node->bounds, r, dir_pos, &tn);
        if (tc) tc->nodes++;
        if (in) {
            if (node->nprims > 0) {
                for (int i = 0; i < node->nprims; ++i) {
                    int o = b->prims[node->offset + i];
                    const TkObject* ob = &s->objects[o];
                    int h;
                    if (ob->kind == 0) h = mesh_intersect(s, o, r, is, faithful, tc);
                    else {
                        if (tc) tc->prims++;
                        h = sphere_intersect(&s->spheres[ob->first], r, is);
                        if (h) is->prim = s->n_tris + ob->first;
                    }
                    if (h) { hit = 1; is->obj = o; }
                }
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            } else {
                int far_first = faithful ? 1 : 0;
                int go_second = dir_pos[node->axis] ? far_first : !far_first;
                if (go_second) { stack[to_visit++] = cur + 1; cur = node->offset; }
                else { stack[to_visit++] = node->offset; cur = cur + 1; }
            }
        } else {
            if (to_visit == 0) break;
            cur = stack[--to_visit];
        }
    }
    return hit;
}

void orc_trace_rays(orc_scene* s, int64_t n, const float* org, const float* dir, const float* tmax,
                    int32_t* prim, float* t, float* p, float* nrm, float* perr, int faithful) {
    for (int64_t i = 0; i < n; ++i) {
        Ray r = make_ray(ld3(org + 3 * i), ld3(dir + 3 * i));
        if (tmax) r.tmax = tmax[i];
        Isect is; memset(&is, 0, sizeof(is)); is.prim = -1;
        int h = scene_intersect(s, &r, &is, faithful, 0);
        prim[i] = h ? is.prim : -1;
        if (t) t[i] = h ? (float)r.tmax : 0.f;
        if (p) st3(p + 3 * i, h ? is.p : V(0, 0, 0));
        if (nrm) st3(nrm + 3 * i, h ? is.n : V(0, 0, 0));
        if (perr) st3(perr + 3 * i, h ? is.perr : V(0, 0, 0));
    }
}

/* ---------------------------------------------------------------- camera */
/* src/Matrix4.cpp:118-128 */
static v3 mat_pos(const float m[16], v3 p) {
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1) return V(xp, yp, zp);
    return vdivs(V(xp, yp, zp), wp);
}
/* src/Matrix4.h:247-253 */
static v3 mat_dir(const float m[16], v3 v) {
    return V(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
             m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
/* src/core/camera.cpp:51-71 with lensRadius = 0 (src/main.cpp:46) */
static Ray camera_ray(const orc_scene* s, float fx, float fy) {
    v3 d = vnormalize(mat_pos(s->r2c, V(fx, fy, 0)));
    v3 o = V(0, 0, 0);
    return make_ray(mat_pos(s->c2w, o), vnormalize(mat_dir(s->c2w, d)));
}
void orc_camera_rays(orc_scene* s, int64_t n, const float* raster, float* org, float* dir) {
    for (int64_t i = 0; i < n; ++i) {
        Ray r = camera_ray(s, raster[2 * i], raster[2 * i + 1]);
        st3(org + 3 * i, r.o); st3(dir + 3 * i, r.d);
    }
}

/* ------------------------------------------------------------- materials */
/* src/core/Material.hpp:77-80 */
static v3 mat_reflect(v3 wo, v3 N) { return vsub(vscale(N, 2 * vdot(wo, N)), wo); }
/* src/core/Material.hpp:82-92 */
static int mat_refract(v3 wi, v3 n, float eta, v3* wt) {
    float cos_i = vdot(wi, n);
    float sin2_i = tk_max(0, 1 - cos_i * cos_i);   /* std::max<float>(0, x) */
    float sin2_t = eta * eta * sin2_i;
    if (sin2_t >= 1) return 0;
    float cos_t = sqrtf(1 - sin2_t);
    *wt = vadd(vscale(vneg(wi), eta), vscale(n, eta * cos_i - cos_t));
    return 1;
}
/* src/core/Material.hpp:94-99 */
static v3 to_world(v3 a, v3 N) {
    v3 B, C;
    coordinate_system(N, &B, &C);
    return vadd(vadd(vscale(B, a.x), vscale(C, a.y)), vscale(N, a.z));
}
/* src/core/Material.cpp:41-61 */
static float fr_dielectric(float cos_i, float eta_i, float eta_t) {
    cos_i = tk_max(tk_min(cos_i, 1.0f), -1.0f);
    int entering = cos_i > 0.f;
    if (!entering) { float t = eta_i; eta_i = eta_t; eta_t = t; cos_i = fabsf(cos_i); }
    float sin_i = sqrtf(tk_max(0.0f, 1 - cos_i * cos_i));
    float sin_t = eta_i / eta_t * sin_i;
    if (sin_t >= 1) return 1;
    float cos_t = sqrtf(tk_max(0.0f, 1 - sin_t * sin_t));
    float rparl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    float rperp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (rparl * rparl + rperp * rperp) / 2;
}
/* DiffuseMaterial::f, src/core/Material.cpp:36-39; mirror and glass return
 * black (src/core/Material.hpp:128, :143). */
static v3 mat_f(const TkMaterial* m, v3 wo, v3 wi, v3 N) {
    if (m->type != TK_DIFFUSE) return V(0, 0, 0);
    return vdot(wo, N) * vdot(wi, N) > 0 ? vscale(ld3(m->R), k_inv_pi) : V(0, 0, 0);
}
/* sample_f of the three materials: src/core/Material.cpp:22-34 (diffuse),
 * :63-88 (glass), :90-95 (mirror).  mode: 0 Radiance, 1 Importance. */
static v3 mat_sample_f(const TkMaterial* m, int mode, v3 wo, v3* wi, v3 N, float ux, float uy, float* pdf) {
    *pdf = 0; *wi = V(0, 0, 0);
    if (m->type == TK_DIFFUSE) {
        v3 tmp = cosine_hemisphere(ux, uy);
        if (tmp.z < 0) tmp.z *= -1;
        *wi = to_world(tmp, N);
        *pdf = vdot(wo, N) * vdot(*wi, N) > 0 ? vdot(*wi, N) * k_inv_pi : 0;
        return mat_f(m, wo, *wi, N);
    } else if (m->type == TK_GLASS) {
        float pr = fr_dielectric(vdot(wo, N), m->etaA, m->etaB);
        if (ux < pr) {
            *wi = mat_reflect(wo, N);
            *pdf = pr;
            return vdivs(vscale(ld3(m->R), pr), vabsdot(*wi, N));
        } else {
            int entering = vdot(wo, N) > 0;
            float etai = entering ? m->etaA : m->etaB;
            float etat = entering ? m->etaB : m->etaA;
            if (!mat_refract(wo, entering ? N : vneg(N), etai / etat, wi)) return V(0, 0, 0);
            v3 ft = vscale(ld3(m->T), 1 - pr);
            if (mode == 0) ft = vscale(ft, (etai * etai) / (etat * etat));
            *pdf = 1 - pr;
            return vdivs(ft, vabsdot(*wi, N));
        }
    } else { /* mirror */
        *wi = mat_reflect(wo, N);
        *pdf = 1;
        return vdivs(ld3(m->R), vabsdot(*wi, N));
    }
}
void orc_sample_f(const TkMaterial* m, int mode, int64_t n, const float* wo, const float* nrm,
                  const float* u, float* wi, float* f, float* pdf) {
    ensure_consts();
    for (int64_t i = 0; i < n; ++i) {
        v3 w; float p;
        v3 fv = mat_sample_f(m, mode, ld3(wo + 3 * i), &w, ld3(nrm + 3 * i), u[2 * i], u[2 * i + 1], &p);
        st3(wi + 3 * i, w); st3(f + 3 * i, fv); pdf[i] = p;
    }
}

/* ---------------------------------------------------------------- lights */
/* src/TkMath.h:298-310 with pred = cdf[idx] <= u (src/core/sampling.h:69) */
static int dist_sample_discrete(const Dist1D* d, float u, float* pdf) {
    long sz = d->n + 1;
    long size = sz - 2, first = 1;
    while (size > 0) {
        long half = size >> 1, middle = first + half;
        int pr = d->cdf[middle] <= u;
        first = pr ? middle + 1 : first;
        size = pr ? size - (half + 1) : half;
    }
    long off = first - 1;
    if (off < 0) off = 0;
    if (off > sz - 2) off = sz - 2;
    if (pdf) *pdf = d->cdf[off + 1] - d->cdf[off];
    return (int)off;
}
/* Triangle::Sample, src/Triangle.cpp:56-66 */
static Isect tri_sample(const Tri* t, float ux, float uy, float* pdf) {
    float su0 = sqrtf(ux);                                /* src/core/sampling.cpp:54-58 */
    float bx = 1 - su0, by = uy * su0;
    Isect r; memset(&r, 0, sizeof(r));
    r.p = vadd(vadd(vscale(t->v0, bx), vscale(t->v1, by)), vscale(t->v2, 1 - bx - by));
    r.n = t->normal;
    v3 s = vadd(vadd(vabs(vscale(t->v0, bx)), vabs(vscale(t->v1, by))), vabs(vscale(t->v2, 1 - bx - by)));
    r.perr = vscale(s, gamma_n(6));
    *pdf = 1 / t->area;
    return r;
}
/* AreaLight::L, src/Light.cpp:90-93 */
static v3 light_L(const TkLight* l, v3 n, v3 w) {
    return (l->two_sided || vdot(n, w) > 0) ? ld3(l->Le) : V(0, 0, 0);
}
/* AreaLight::sample_Le, src/Light.cpp:41-75, through MeshTriangle::Sample
 * (src/Triangle.cpp:190-197) whose triangle pick is a get_random_float(). */
static v3 light_sample_le(const orc_scene* s, int li, float u1x, float u1y, float u2x, float u2y,
                          float upick, Ray* ray, v3* nlight, float* pdf_pos, float* pdf_dir) {
    const TkLight* l = &s->lights[li];
    const TkObject* ob = &s->objects[l->object];
    float pd, tmp;
    int idx = dist_sample_discrete(&s->light_dist[li], upick, &pd);
    Isect ps = tri_sample(&s->tris[ob->first + idx], u1x, u1y, &tmp);
    *pdf_pos = pd * tmp;
    *nlight = ps.n;
    v3 w;
    if (l->two_sided) {
        float ux = u2x;
        if (ux < 0.5) { ux = tk_min(ux * 2, k_one_minus_eps); w = cosine_hemisphere(ux, u2y); }
        else { ux = tk_min((ux - 0.5f) * 2, k_one_minus_eps); w = cosine_hemisphere(ux, u2y); w.z *= -1; }
        *pdf_dir = 0.5f * (fabsf(w.z) * k_inv_pi);
    } else {
        w = cosine_hemisphere(u2x, u2y);
        *pdf_dir = w.z * k_inv_pi;
    }
    v3 v1, v2, n = ps.n;
    coordinate_system(n, &v1, &v2);
    w = vadd(vadd(vscale(v1, w.x), vscale(v2, w.y)), vscale(n, w.z));
    *ray = spawn_ray(&ps, w);
    return light_L(l, ps.n, w);
}
void orc_sample_le(orc_scene* s, int li, int64_t n, const float* u1, const float* u2,
                   const float* upick, float* org, float* dir, float* nrm, float* le,
                   float* pdf_pos, float* pdf_dir) {
    for (int64_t i = 0; i < n; ++i) {
        Ray r; v3 nl; float pp, pd;
        v3 L = light_sample_le(s, li, u1[2 * i], u1[2 * i + 1], u2[2 * i], u2[2 * i + 1], upick[i], &r, &nl, &pp, &pd);
        st3(org + 3 * i, r.o); st3(dir + 3 * i, r.d); st3(nrm + 3 * i, nl); st3(le + 3 * i, L);
        pdf_pos[i] = pp; pdf_dir[i] = pd;
    }
}
/* AreaLight::sample_Li (src/Light.cpp:24-33) through MeshTriangle::Sample
 * (src/Triangle.cpp:199-206) and Triangle::Sample (src/Triangle.cpp:68-81). */
static v3 light_sample_li(const orc_scene* s, int li, const Isect* ref, float u0, float ux, float uy,
                          Isect* it, v3* wi, float* pdf) {
    const TkLight* l = &s->lights[li];
    const TkObject* ob = &s->objects[l->object];
    float pd, tmp;
    int idx = dist_sample_discrete(&s->light_dist[li], u0, &pd);
    Isect ps = tri_sample(&s->tris[ob->first + idx], ux, uy, &tmp);
    v3 w = vsub(ps.p, ref->p);
    float dist2 = vdot(w, w);
    if (dist2 == 0) tmp = 0;
    else { w = vnormalize(w); tmp *= dist2 / vabsdot(ps.n, vneg(w)); }
    *pdf = pd * tmp;
    v3 dp = vsub(ps.p, ref->p);
    if (*pdf == 0 || vdot(dp, dp) == 0) return V(0, 0, 0);
    *wi = vnormalize(vsub(ps.p, ref->p));
    *it = ps;
    return light_L(l, ps.n, vneg(*wi));
}

/* --------------------------------------------------------------- sampler */
typedef struct {
    int px, py, sample_idx, spp, xs, ys, dim, seed;   /* src/core/sampler.h:10-54 */
    uint32_t rng_seed, rng_pixel, rng_iter;
    int rng_cost;
    int no_jitter;   /* StratifiedSampler(..., jitter = false): delta = 0.5 (src/core/sampler.cpp:51, :61-62) */
} Sampler;
static float sampler_jitter(Sampler* s, int slot) {
    if (s->no_jitter) return 0.5f;
    if (s->rng_cost) g_mt_sink = mt_cost(s->rng_pixel + (uint32_t)slot);
    return orc_rng_u01(s->rng_seed, s->rng_pixel, s->rng_iter, (uint32_t)slot);
}
/* src/core/sampler.cpp:46-53 */
static float sampler_get1d(Sampler* s) {
    uint64_t hash = orc_hash_pixel(s->px, s->py, s->dim, s->seed);
    int stratum = orc_permutation_element((uint32_t)s->sample_idx, (uint32_t)s->spp, (uint32_t)hash);
    int d = s->dim;
    ++s->dim;
    float delta = sampler_jitter(s, 2 * d);
    return (stratum + delta) / s->spp;
}
/* src/core/sampler.cpp:55-64 */
static void sampler_get2d(Sampler* s, float* ox, float* oy) {
    uint64_t hash = orc_hash_pixel(s->px, s->py, s->dim, s->seed);
    int stratum = orc_permutation_element((uint32_t)s->sample_idx, (uint32_t)s->spp, (uint32_t)hash);
    int d = s->dim;
    s->dim += 2;
    int x = stratum % s->xs, y = stratum / s->xs;
    float dx = sampler_jitter(s, 2 * d);
    float dy = sampler_jitter(s, 2 * d + 1);
    *ox = (x + dx) / s->xs;
    *oy = (y + dy) / s->ys;
}

/* ------------------------------------------------------------ photon map */
typedef struct {
    v3 pos, power;
    uint8_t theta, phi;
    int16_t split;
} Photon;                                                /* src/tracer/neighbor.h:9-18 */
typedef struct { int32_t pt, left, right; } KdNode;      /* src/tracer/neighbor.h:20-26 (indices) */

struct orc_map {
    Photon* ph; int64_t n, cap;
    KdNode* nodes; int64_t n_nodes;
    int k;
    float cos_t[256], sin_t[256], cos_p[256], sin_p[256];
    box3 bbox;           /* stays empty in split_mode 0 (Union result discarded) */
    int built;
    int tie_by_index;    /* among photons tied at the k-th distance keep the lowest indices (orc_map_set_tie_rule) */
};

/* src/tracer/Photonmap.cpp:26-33 */
void orc_direction_tables(float* ct, float* st, float* cp, float* sp) {
    ensure_consts();
    for (int i = 0; i < 256; ++i) {
        double radians = (double)i * k_pi / 255.0;
        ct[i] = (float)cos(radians); st[i] = (float)sin(radians);
        cp[i] = (float)cos(2.0 * radians); sp[i] = (float)sin(2.0 * radians);
    }
}
/* src/tracer/TkSppmTracer.cpp:135-140: float acos / atan2 and float products
 * (wo.z, Math::inv_pi and Math::pi are floats, 255 converts to float). */
void orc_encode_direction(const float wo[3], uint8_t* theta, uint8_t* phi) {
    ensure_consts();
    const float inv_pi = (float)k_inv_pi, pi = (float)k_pi;
    float th = 255 * tk_acosf(wo[2]) * inv_pi;
    int t = th != th ? 255 : (int)th;          /* int(NaN) is undefined in the reference; clamp */
    if (t > 255) t = 255;
    if (t < 0) t = 0;
    int p = (int)(255 * tk_atan2f(wo[1], wo[0]) / (2 * pi));
    if (p < 0) p += 255;
    if (p > 255) p = 255;
    *theta = (uint8_t)t; *phi = (uint8_t)p;
}

orc_map* orc_map_create(int64_t capacity, int k) {
    ensure_consts();
    orc_map* m = (orc_map*)calloc(1, sizeof(orc_map));
    m->cap = capacity; m->k = k;
    m->ph = (Photon*)malloc(sizeof(Photon) * (size_t)(capacity ? capacity : 1));
    m->nodes = (KdNode*)malloc(sizeof(KdNode) * (size_t)(capacity ? capacity : 1));
    orc_direction_tables(m->cos_t, m->sin_t, m->cos_p, m->sin_p);
    m->bbox = box_empty();
    return m;
}
void orc_map_destroy(orc_map* m) { if (m) { free(m->ph); free(m->nodes); free(m); } }
void orc_map_reset(orc_map* m) { m->n = 0; m->built = 0; }   /* src/tracer/Photonmap.h:34 */

/* storePhoton, src/tracer/Photonmap.cpp:44-53 */
static int map_store_one(orc_map* m, v3 pos, v3 power, uint8_t theta, uint8_t phi) {
    if (m->n >= m->cap) return 0;
    Photon* p = &m->ph[m->n++];
    p->pos = pos; p->power = power; p->theta = theta; p->phi = phi; p->split = 4;
    return 1;
}
int64_t orc_map_store(orc_map* m, const float* pos, const float* power, const uint8_t* theta,
                      const uint8_t* phi, int64_t n) {
    int64_t s = 0;
    for (int64_t i = 0; i < n; ++i)
        s += map_store_one(m, ld3(pos + 3 * i), ld3(power + 3 * i), theta[i], phi[i]);
    return s;
}

/* Selection of the element of rank `want` in idx[lo,hi) by pos[dim]
 * (quick_median, src/tracer/quickselect.h:114-120; any correct selection
 * yields the same node photon unless coordinates tie). */
static void kd_select(const Photon* ph, int32_t* idx, int64_t lo, int64_t hi, int64_t want, int dim) {
    while (hi - lo > 1) {
        int64_t mid = lo + (hi - lo) / 2;
        float a = vget(ph[idx[lo]].pos, dim), b = vget(ph[idx[mid]].pos, dim), c = vget(ph[idx[hi - 1]].pos, dim);
        float pivot = (a < b) ? ((b < c) ? b : (a < c ? c : a)) : ((a < c) ? a : (b < c ? c : b));
        int64_t i = lo, j = hi - 1;
        while (i <= j) {
            while (vget(ph[idx[i]].pos, dim) < pivot) ++i;
            while (vget(ph[idx[j]].pos, dim) > pivot) --j;
            if (i <= j) { int32_t t = idx[i]; idx[i] = idx[j]; idx[j] = t; ++i; --j; }
        }
        if (want <= j) hi = j + 1;
        else if (want >= i) lo = i;
        else return;
    }
}

/* balance, src/tracer/Photonmap.cpp:132-179.  In split_mode 0 dim is always 2:
 * storePhoton discards the result of Union(bbox,pos) (src/tracer/Photonmap.cpp:50)
 * so bbox keeps pMin=+inf,pMax=-inf and maxExtent() returns 2. */
static int32_t kd_balance(orc_map* m, int32_t* idx, int64_t start, int64_t end, int64_t* nidx, int split_mode, box3 bb) {
    int32_t my = (int32_t)(*nidx);
    KdNode* node = &m->nodes[my];
    node->left = node->right = -1;
    int dim = 2;
    if (split_mode == 1) dim = box_max_extent(bb);
    int64_t median = start + (end - start) / 2;
    kd_select(m->ph, idx, start, end, median, dim);
    m->ph[idx[median]].split = (int16_t)dim;
    node->pt = idx[median];
    float mv = vget(m->ph[idx[median]].pos, dim);
    if (median > start) {
        ++*nidx;
        if (start < median - 1) {
            box3 lb = bb; (&lb.mx.x)[dim] = mv;
            int32_t c = kd_balance(m, idx, start, median, nidx, split_mode, lb);
            m->nodes[my].left = c;
        } else {
            KdNode* child = &m->nodes[*nidx];
            child->pt = idx[start]; child->left = child->right = -1;
            m->nodes[my].left = (int32_t)(*nidx);
        }
    }
    if (median + 1 != end) {
        ++*nidx;
        if (median + 2 != end) {
            box3 rb = bb; (&rb.mn.x)[dim] = mv;
            int32_t c = kd_balance(m, idx, median + 1, end, nidx, split_mode, rb);
            m->nodes[my].right = c;
        } else {
            KdNode* child = &m->nodes[*nidx];
            child->pt = idx[median + 1]; child->left = child->right = -1;
            m->nodes[my].right = (int32_t)(*nidx);
        }
    }
    return my;
}
/* buildKdTree, src/tracer/Photonmap.cpp:96-103 */
void orc_map_build(orc_map* m, int split_mode) {
    m->built = 0;
    if (m->n == 0) return;
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)m->n);
    box3 bb = box_empty();
    for (int64_t i = 0; i < m->n; ++i) { idx[i] = (int32_t)i; m->ph[i].split = 4; bb = box_union_p(bb, m->ph[i].pos); }
    int64_t nidx = 0;
    kd_balance(m, idx, 0, m->n, &nidx, split_mode, bb);
    m->n_nodes = nidx + 1;
    free(idx);
    m->built = 1;
}
int64_t orc_map_split_histogram(orc_map* m, int64_t hist[5]) {
    for (int i = 0; i < 5; ++i) hist[i] = 0;
    for (int64_t i = 0; i < m->n; ++i) hist[m->ph[i].split]++;
    return m->n;
}

typedef struct {
    int found; v3 dst; float* dist2; int32_t* nb;
} Nearest;                                               /* src/tracer/neighbor.h:28-35 */

/* neighbor_search, src/tracer/neighbor.cpp:5-85 */
static void neighbor_search(const orc_map* m, Nearest* np, int32_t ni, int k) {
    if (ni < 0) return;
    const KdNode* node = &m->nodes[ni];
    const Photon* p = &m->ph[node->pt];
    v3 np_pos = p->pos;
    if (p->split < 3) {
        float dist = vget(np->dst, p->split) - vget(np_pos, p->split);
        if (dist > 0) {
            neighbor_search(m, np, node->right, k);
            if (dist * dist < np->dist2[k]) neighbor_search(m, np, node->left, k);
        } else {
            neighbor_search(m, np, node->left, k);
            if (dist * dist < np->dist2[k]) neighbor_search(m, np, node->right, k);
        }
    }
    v3 dv = vsub(np_pos, np->dst);
    float dist2 = vdot(dv, dv);
    if (dist2 < np->dist2[k]) {
        if (np->found < k) {
            np->dist2[np->found] = dist2;
            np->nb[np->found] = node->pt;
            np->found++;
            if (np->found == k) {
                int half = (np->found - 2) >> 1;
                for (int kk = half; kk >= 0; --kk) {
                    int parent = kk;
                    int32_t pp = np->nb[kk];
                    float pd = np->dist2[kk];
                    while (parent <= half) {
                        int child = 2 * parent + 1;
                        if ((child + 1) < np->found && np->dist2[child] < np->dist2[child + 1]) child++;
                        if (pd > np->dist2[child]) break;
                        np->dist2[parent] = np->dist2[child];
                        np->nb[parent] = np->nb[child];
                        parent = child;
                    }
                    np->dist2[parent] = pd;
                    np->nb[parent] = pp;
                }
                np->dist2[k] = np->dist2[0];
            }
        } else {
            int half = (np->found - 2) >> 1;
            int parent = 0, child = 1;
            while (parent <= half) {
                if ((child + 1) < np->found && np->dist2[child] < np->dist2[child + 1]) child++;
                if (dist2 > np->dist2[child]) break;
                np->dist2[parent] = np->dist2[child];
                np->nb[parent] = np->nb[child];
                parent = child;
                child += child + 1;
            }
            np->dist2[parent] = dist2;
            np->nb[parent] = node->pt;
            np->dist2[k] = np->dist2[0];
        }
    }
}

/* Sum of radiance_estimate over a found set, src/tracer/Photonmap.cpp:118-127 */
static v3 estimate_sum(const orc_map* m, const int32_t* nb, int found, v3 n, v3 wo, v3 R) {
    v3 ret = V(0, 0, 0);
    for (int i = 0; i < found; ++i) {
        const Photon* p = &m->ph[nb[i]];
        float tmp = m->sin_t[p->theta];
        v3 wi = V(tmp * m->cos_p[p->phi], tmp * m->sin_p[p->phi], m->cos_t[p->theta]);
        v3 f = vdot(wo, n) * vdot(wi, n) > 0 ? vscale(R, k_inv_pi) : V(0, 0, 0);
        ret = vadd(ret, vmul(vscale(p->power, vabsdot(wi, n)), f));
    }
    return ret;
}
static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

/* radiance_estimate, src/tracer/Photonmap.cpp:105-130 */
static v3 map_radiance_estimate(const orc_map* m, v3 p, v3 n, v3 wo, v3 R, float* radius2, int* found,
                                float* d2buf, int32_t* nbbuf) {
    Nearest np;
    np.dst = p; np.found = 0; np.dist2 = d2buf; np.nb = nbbuf;
    np.dist2[m->k] = *radius2;
    if (m->built) neighbor_search(m, &np, 0, m->k);
    *radius2 = np.dist2[m->k];
    *found = np.found;
    if (m->tie_by_index && np.found == m->k) {
        /* The search keeps whichever of the photons tied at the k-th distance it met first; the
         * contract of the GPU path is the lowest indices.  Same count, same radius, other members. */
        const float kth = np.dist2[m->k];
        int less = 0;
        for (int i = 0; i < np.found; ++i) {
            v3 dv = vsub(m->ph[np.nb[i]].pos, p);
            if (vdot(dv, dv) < kth) np.nb[less++] = np.nb[i];
        }
        for (int64_t i = 0; i < m->n && less < m->k; ++i) {
            v3 dv = vsub(m->ph[i].pos, p);
            if (vdot(dv, dv) == kth) np.nb[less++] = (int32_t)i;
        }
    }
    return estimate_sum(m, np.nb, np.found, n, wo, R);
}
void orc_map_set_tie_rule(orc_map* m, int by_index) { m->tie_by_index = by_index; }
void orc_map_gather(orc_map* m, int64_t nq, const float* p, const float* n, const float* wo,
                    const float* R, const float* radius2, float* flux, int32_t* found,
                    float* radius2_out, int32_t* idx) {
    float* d2 = (float*)malloc(sizeof(float) * (size_t)(m->k + 1));
    int32_t* nb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m->k + 1));
    for (int64_t q = 0; q < nq; ++q) {
        float r2 = radius2[q]; int fnd;
        v3 f = map_radiance_estimate(m, ld3(p + 3 * q), ld3(n + 3 * q), ld3(wo + 3 * q), ld3(R + 3 * q), &r2, &fnd, d2, nb);
        st3(flux + 3 * q, f); found[q] = fnd; radius2_out[q] = r2;
        if (idx) {
            qsort(nb, (size_t)fnd, sizeof(int32_t), cmp_i32);
            for (int i = 0; i < m->k; ++i) idx[q * m->k + i] = i < fnd ? nb[i] : -1;
        }
    }
    free(d2); free(nb);
}

typedef struct { float d2; int32_t idx; } Cand;
static int cmp_cand(const void* a, const void* b) {
    const Cand* x = (const Cand*)a; const Cand* y = (const Cand*)b;
    if (x->d2 < y->d2) return -1;
    if (x->d2 > y->d2) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
void orc_map_gather_brute(orc_map* m, int64_t nq, const float* p, const float* n, const float* wo,
                          const float* R, const float* radius2, float* flux, int32_t* found,
                          float* radius2_out, int32_t* idx) {
    Cand* c = (Cand*)malloc(sizeof(Cand) * (size_t)(m->n ? m->n : 1));
    int32_t* nb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m->k + 1));
    for (int64_t q = 0; q < nq; ++q) {
        v3 dst = ld3(p + 3 * q);
        int64_t nc = 0;
        for (int64_t i = 0; i < m->n; ++i) {
            v3 dv = vsub(m->ph[i].pos, dst);
            float d2 = vdot(dv, dv);
            if (d2 < radius2[q]) { c[nc].d2 = d2; c[nc].idx = (int32_t)i; ++nc; }
        }
        qsort(c, (size_t)nc, sizeof(Cand), cmp_cand);
        int fnd = nc < m->k ? (int)nc : m->k;
        for (int i = 0; i < fnd; ++i) nb[i] = c[i].idx;
        radius2_out[q] = (nc >= m->k) ? c[m->k - 1].d2 : radius2[q];
        found[q] = fnd;
        qsort(nb, (size_t)fnd, sizeof(int32_t), cmp_i32);
        st3(flux + 3 * q, estimate_sum(m, nb, fnd, ld3(n + 3 * q), ld3(wo + 3 * q), ld3(R + 3 * q)));
        if (idx) for (int i = 0; i < m->k; ++i) idx[q * m->k + i] = i < fnd ? nb[i] : -1;
    }
    free(c); free(nb);
}

/* ---------------------------------------------- pixel update and resolve */
/* src/tracer/TkSppmTracer.cpp:327-340 and :346-359 (same rule for both maps) */
static void pixel_update_one(float flux[3], float* photons, float* radius2, v3 lindir, int found, float r2ret) {
    if (*photons == 0) {
        *radius2 = r2ret;
        *photons = (float)found;
        flux[0] = lindir.x; flux[1] = lindir.y; flux[2] = lindir.z;
    } else {
        float new_photons = (float)(*photons + 0.7 * found);
        float new_radius2 = *radius2 * new_photons / (*photons + found);
        flux[0] = (flux[0] + lindir.x) * new_radius2 / *radius2;
        flux[1] = (flux[1] + lindir.y) * new_radius2 / *radius2;
        flux[2] = (flux[2] + lindir.z) * new_radius2 / *radius2;
        *photons = new_photons;
        *radius2 = new_radius2;
    }
}
void orc_pixel_update(int64_t n, float* flux, float* photons, float* radius2, const float* lindir,
                      const int32_t* found, const float* radius2_ret) {
    for (int64_t i = 0; i < n; ++i)
        pixel_update_one(flux + 3 * i, photons + i, radius2 + i, ld3(lindir + 3 * i), found[i], radius2_ret[i]);
}

/* src/tracer/TkSppmTracer.cpp:514-529.  The reference keeps the two light-path
 * totals in s32 members (src/tracer/TkSppmTracer.h:31-32), which wrap on long
 * jobs; here they are 64-bit and convert to float directly. */
void orc_resolve_rgb8(const TkSPPMPixel* pp, int w, int h, int num_recurse, int64_t shoot_global,
                      int64_t shoot_caustic, int caustic_stored, uint8_t* out) {
    ensure_consts();
    size_t off = 0;
    float sg = (float)shoot_global, sc = (float)shoot_caustic;
    for (int y = h; y-- > 0;) {
        for (int x = 0; x < w; ++x) {
            const TkSPPMPixel* p = &pp[x + y * w];
            float c[3];
            for (int ch = 0; ch < 3; ++ch) {
                float v = p->Ld[ch] / num_recurse;
                if (caustic_stored > 0) v = v + p->causticFlux[ch] / (k_pi * p->causticRadius2 * sc);
                else v = v + 0.f;
                v = v + p->globalFlux[ch] / (k_pi * p->globalRadius2 * sg);
                c[ch] = v;
            }
            for (int ch = 0; ch < 3; ++ch) {
                float cl = tk_max(tk_min(c[ch], 1.0f), 0.0f);
                out[off++] = (uint8_t)(255 * powf(cl, 0.6f));
            }
        }
    }
}

/* ------------------------------------------------------------ SPPM state */
typedef struct { v3 pos, power; uint8_t theta, phi; } PhotonRec;

struct orc_sppm {
    orc_scene* sc;
    TkSppmParam prm;
    int num_recurse, xs, ys, max_depth;
    float rr;
    uint32_t seed;
    int rank, world, threads, split_mode, faithful, rng_cost, no_jitter;
    orc_map* gmap; orc_map* cmap;
    int64_t quota_g, quota_c;
    /* this rank's traced photons (before exchange) */
    PhotonRec* loc_g; int64_t n_loc_g; PhotonRec* loc_c; int64_t n_loc_c;
    int64_t local_shot_g, local_shot_c;     /* cumulative local path counters */
    int64_t tc_photon[3], tc_camera[3];     /* rays, box tests, primitive tests of the passes so far */
    int64_t caustic_hoff;                   /* global paths shot before the current iteration */
    int64_t last_shot_g, last_shot_c;       /* paths of the last pass */
    int64_t shoot_g, shoot_c;               /* cumulative over all ranks */
    int64_t stored_c;
    TkSPPMPixel* pp;
    double yield_g, yield_c;
};

static float photon_rng(const orc_sppm* r, uint64_t H, int kind, int bounce, int slot) {
    uint32_t b = (uint32_t)kind | ((uint32_t)bounce << 4) | ((uint32_t)(H >> 32) << 16);
    if (r->rng_cost) g_mt_sink = mt_cost((uint32_t)H + (uint32_t)slot);
    return orc_rng_u01(r->seed, (uint32_t)H, b, (uint32_t)slot);
}

/* The numbers light path H of `kind` (0 global, 1 caustic, 2 hash-grid variant) draws where the reference
 * calls get_random_float(), in the reference's call order: the triangle pick inside AreaLight::sample_Le
 * (src/Triangle.cpp:192), then the Russian-roulette number of bounce 0, 1, ... (src/tracer/TkSppmTracer.cpp:144,
 * :204; src/TkSPPM.cpp:236).  Row i of out (per numbers) is path H0 + i.  Fed to the reference's own loops they
 * make it trace the paths this port (and the HIP kernels) trace for the same indices. */
void orc_photon_draws(uint32_t seed, int kind, uint64_t H0, int64_t n, int per, float* out) {
    orc_sppm r; memset(&r, 0, sizeof(r));
    r.seed = seed;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t H = H0 + (uint64_t)i;
        float* row = out + i * per;
        if (per > 0) row[0] = photon_rng(&r, H, kind, 0, 0);
        for (int b = 0; b + 1 < per; ++b) row[1 + b] = photon_rng(&r, H, kind, b, 1);
    }
}
/* Likewise the roulette numbers of PathTracer::Li (src/tracer/TkPathTracer.cpp:165) for sample `sample` of
 * pixels 0 .. n-1: row p holds the numbers of bounce 0 .. per-1. */
void orc_pt_draws(uint32_t seed, int sample, int64_t n, int per, float* out) {
    for (int64_t p = 0; p < n; ++p)
        for (int b = 0; b < per; ++b)
            out[p * per + b] = orc_rng_u01(seed, (uint32_t)p, (uint32_t)sample, 0x10000u + (uint32_t)b);
}

/* tracePhoton (kind 0, src/tracer/TkSppmTracer.cpp:104-157) and
 * traceCausticPhoton (kind 1, :159-217).  Photons are returned in recs instead
 * of being pushed into the shared map; the caller appends them in path order. */
static int trace_photon_path(const orc_sppm* r, uint64_t H, int kind, PhotonRec* recs, TraceCounters* tc) {
    const orc_scene* s = r->sc;
    int nrec = 0, dim = 0;
    int nl = s->n_lights;
    if (nl == 0) return 0;
    int li = (int)(orc_radical_inverse(dim++, H) * nl);
    if (li > nl - 1) li = nl - 1;
    float pdf = 1.0f / nl;
    float u0x = orc_radical_inverse(dim, H), u0y = orc_radical_inverse(dim + 1, H);
    float u1x = orc_radical_inverse(dim + 2, H), u1y = orc_radical_inverse(dim + 3, H);
    dim += 4;
    Ray ray; v3 normal; float pos_pdf, dir_pdf;
    float upick = photon_rng(r, H, kind, 0, 0);
    v3 power = light_sample_le(s, li, u0x, u0y, u1x, u1y, upick, &ray, &normal, &pos_pdf, &dir_pdf);
    if (pos_pdf == 0 || dir_pdf == 0) return 0;
    power = vdivs(vscale(power, vabsdot(ray.d, normal)), pos_pdf * dir_pdf * pdf);
    int has_glossy = 0;
    int depth = kind == 0 ? r->max_depth : 64;
    for (int i = 0; i < depth; ++i) {
        Isect is; memset(&is, 0, sizeof(is));
        if (!scene_intersect(s, &ray, &is, r->faithful, tc)) break;
        v3 wo = vneg(ray.d);
        const TkMaterial* m = &s->mats[s->objects[is.obj].material];
        int diffuse = m->type == TK_DIFFUSE;
        if (kind == 0) {
            if (diffuse) {
                PhotonRec* pr = &recs[nrec++];
                pr->pos = is.p; pr->power = power;
                float w3[3] = {wo.x, wo.y, wo.z};
                orc_encode_direction(w3, &pr->theta, &pr->phi);
            }
        } else {
            has_glossy |= !diffuse;
            if (diffuse) {
                if (has_glossy) {
                    PhotonRec* pr = &recs[nrec++];
                    pr->pos = is.p; pr->power = power;
                    float w3[3] = {wo.x, wo.y, wo.z};
                    orc_encode_direction(w3, &pr->theta, &pr->phi);
                }
                break;
            }
        }
        if (photon_rng(r, H, kind, i, 1) > r->rr) break;
        v3 wi; float bpdf = 0;
        float ux = orc_radical_inverse(dim, H), uy = orc_radical_inverse(dim + 1, H);
        v3 f = mat_sample_f(m, 1, wo, &wi, is.n, ux, uy, &bpdf);
        if (bpdf == 0 || (f.x == 0 && f.y == 0 && f.z == 0)) break;
        power = vdivs(vscale(vmul(power, f), vabsdot(wi, is.n)), bpdf * r->rr);
        ray = spawn_ray(&is, wi);
    }
    return nrec;
}

/* ------------------------------------------------------- parallel helpers */
typedef void (*job_fn)(void* arg, int64_t i, int tid);
typedef struct { job_fn fn; void* arg; int64_t n; int64_t chunk; int64_t next; pthread_mutex_t mu; } JobPool;
typedef struct { JobPool* pool; int tid; } JobThread;
static void* job_worker(void* a) {
    JobThread* jt = (JobThread*)a;
    JobPool* p = jt->pool;
    while (1) {
        pthread_mutex_lock(&p->mu);
        int64_t b = p->next; p->next += p->chunk;
        pthread_mutex_unlock(&p->mu);
        if (b >= p->n) break;
        int64_t e = b + p->chunk < p->n ? b + p->chunk : p->n;
        for (int64_t i = b; i < e; ++i) p->fn(p->arg, i, jt->tid);
    }
    return 0;
}
static void parallel_for(int threads, int64_t n, int64_t chunk, job_fn fn, void* arg) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    JobPool pool; pool.fn = fn; pool.arg = arg; pool.n = n; pool.chunk = chunk > 0 ? chunk : 1; pool.next = 0;
    pthread_mutex_init(&pool.mu, 0);
    pthread_t th[256]; JobThread jt[256];
    for (int t = 1; t < threads; ++t) { jt[t].pool = &pool; jt[t].tid = t; pthread_create(&th[t], 0, job_worker, &jt[t]); }
    jt[0].pool = &pool; jt[0].tid = 0;
    job_worker(&jt[0]);
    for (int t = 1; t < threads; ++t) pthread_join(th[t], 0);
    pthread_mutex_destroy(&pool.mu);
}

/* ----------------------------------------------------------- photon pass */
typedef struct {
    const orc_sppm* r; int kind; int64_t m0; int64_t halton_off; int slots;
    PhotonRec* stage; int32_t* counts; TraceCounters* tcs;
} PhotonJob;
static void photon_job(void* a, int64_t i, int tid) {
    PhotonJob* j = (PhotonJob*)a;
    const orc_sppm* r = j->r;
    /* local path index m -> Halton index (m + offset) * world + rank */
    uint64_t H = (uint64_t)(j->m0 + i + j->halton_off) * (uint64_t)r->world + (uint64_t)r->rank;
    j->counts[i] = trace_photon_path(r, H, j->kind, j->stage + i * j->slots, &j->tcs[tid]);
}

static void photon_pass_kind(orc_sppm* r, int kind, TkIterStats* st) {
    int64_t quota = kind == 0 ? r->quota_g : r->quota_c;
    PhotonRec* out = kind == 0 ? r->loc_g : r->loc_c;
    int64_t stored = 0, shot = 0;
    int64_t m0 = kind == 0 ? r->local_shot_g : r->local_shot_c;
    /* caustic Halton index = j + shootGlobalPhotons (src/tracer/TkSppmTracer.cpp:419), with the
     * global count taken at the start of the iteration so that the two passes of an iteration
     * do not depend on each other */
    int64_t hoff = kind == 0 ? 0 : r->caustic_hoff;
    int slots = kind == 0 ? r->max_depth : 1;
    double* yield = kind == 0 ? &r->yield_g : &r->yield_c;
    TraceCounters tcs[256]; memset(tcs, 0, sizeof(tcs));
    int64_t emitted_cap = quota * 4000 + 100000;  /* stop if nothing can be stored */
    while (stored < quota && shot < emitted_cap) {
        int64_t need = quota - stored;
        int64_t B = (int64_t)(need / *yield * 1.02) + 256;
        if (B > 4000000) B = 4000000;
        PhotonRec* stage = (PhotonRec*)malloc(sizeof(PhotonRec) * (size_t)B * (size_t)slots);
        int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * (size_t)B);
        PhotonJob job = {r, kind, m0 + shot, hoff, slots, stage, counts, tcs};
        parallel_for(r->threads, B, 64, photon_job, &job);
        int64_t used = 0, got = 0;
        for (int64_t i = 0; i < B && stored < quota; ++i) {
            for (int c = 0; c < counts[i] && stored < quota; ++c) { out[stored++] = stage[i * slots + c]; ++got; }
            used = i + 1;
        }
        shot += used;
        if (used > 0 && got > 0) *yield = 0.5 * *yield + 0.5 * ((double)got / (double)used);
        else if (got == 0) *yield *= 0.25;
        if (*yield < 1e-4) *yield = 1e-4;
        free(stage); free(counts);
    }
    if (kind == 0) { r->n_loc_g = stored; r->last_shot_g = shot; r->local_shot_g += shot; }
    else { r->n_loc_c = stored; r->last_shot_c = shot; r->local_shot_c += shot; }
    if (st) for (int t = 0; t < 256; ++t) st->rays += tcs[t].rays;
    for (int t = 0; t < 256; ++t) { r->tc_photon[0] += tcs[t].rays; r->tc_photon[1] += tcs[t].nodes; r->tc_photon[2] += tcs[t].prims; }
}

static void install_map(orc_map* m, const PhotonRec* recs, int64_t n) {
    orc_map_reset(m);
    for (int64_t i = 0; i < n; ++i) map_store_one(m, recs[i].pos, recs[i].power, recs[i].theta, recs[i].phi);
}

void orc_sppm_photon_pass(orc_sppm* r, int iter, TkIterStats* st) {
    (void)iter;
    r->caustic_hoff = r->local_shot_g;
    photon_pass_kind(r, 0, st);
    if (r->quota_c > 0) photon_pass_kind(r, 1, st); else { r->n_loc_c = 0; r->last_shot_c = 0; }
    if (st) {
        st->global_paths = r->last_shot_g; st->caustic_paths = r->last_shot_c;
    }
    if (r->world == 1) {
        install_map(r->gmap, r->loc_g, r->n_loc_g);
        install_map(r->cmap, r->loc_c, r->n_loc_c);
        r->shoot_g += r->last_shot_g; r->shoot_c += r->last_shot_c;
        r->stored_c = r->n_loc_c;
    }
    if (st) {
        st->global_stored = r->n_loc_g; st->caustic_stored = r->n_loc_c;
        st->shoot_global = r->shoot_g; st->shoot_caustic = r->shoot_c;
    }
}

int64_t orc_sppm_read_photons(orc_sppm* r, int which, float* pos, float* power, uint8_t* theta,
                              uint8_t* phi, int64_t* shot_local) {
    const PhotonRec* recs = which == 0 ? r->loc_g : r->loc_c;
    int64_t n = which == 0 ? r->n_loc_g : r->n_loc_c;
    for (int64_t i = 0; i < n; ++i) {
        if (pos) st3(pos + 3 * i, recs[i].pos);
        if (power) st3(power + 3 * i, recs[i].power);
        if (theta) theta[i] = recs[i].theta;
        if (phi) phi[i] = recs[i].phi;
    }
    if (shot_local) *shot_local = which == 0 ? r->last_shot_g : r->last_shot_c;
    return n;
}
void orc_sppm_set_map_photons(orc_sppm* r, int which, const float* pos, const float* power,
                              const uint8_t* theta, const uint8_t* phi, int64_t n, int64_t shot_total) {
    orc_map* m = which == 0 ? r->gmap : r->cmap;
    orc_map_reset(m);
    orc_map_store(m, pos, power, theta, phi, n);
    if (which == 0) r->shoot_g += shot_total; else { r->shoot_c += shot_total; r->stored_c = n; }
}

void orc_sppm_build_maps(orc_sppm* r, TkIterStats* st) {
    (void)st;
    orc_map_build(r->gmap, r->split_mode);
    if (r->cmap->n > 0) orc_map_build(r->cmap, r->split_mode);
}

/* ----------------------------------------------------------- camera pass */
typedef struct { int64_t rays, gg, gc, gphot; TraceCounters tc; float* d2; int32_t* nb; } CamThread;

/* Scene::uniformSampleOneLight, src/core/Scene.cpp:43-61, with the four numbers it takes from the
 * sampler given: which light, the point on it (2), which triangle of its mesh. */
static v3 one_light_with(const orc_scene* s, int faithful, const Isect* it, float u_choice, float ulx, float uly,
                         float u0, TraceCounters* tc) {
    int nl = s->n_lights;
    int idx = (int)(u_choice * nl);
    if (idx > nl - 1) idx = nl - 1;
    float pdf_choice = 1.0f / nl;
    v3 ws = V(0, 0, 0); float pdf_light = 0; Isect li; memset(&li, 0, sizeof(li));
    v3 Li = light_sample_li(s, idx, it, u0, ulx, uly, &li, &ws, &pdf_light);
    if (Li.x == 0 && Li.y == 0 && Li.z == 0) return V(0, 0, 0);
    Ray sr = spawn_ray_to(it, &li);
    Isect tmp; memset(&tmp, 0, sizeof(tmp));
    if (scene_intersect(s, &sr, &tmp, faithful, tc)) return V(0, 0, 0);
    const TkMaterial* m = &s->mats[s->objects[it->obj].material];
    v3 f = mat_f(m, it->wo, ws, it->n);
    return vdivs(vscale(vmul(Li, f), vabsdot(ws, it->n)), pdf_light * pdf_choice);
}
static v3 uniform_sample_one_light(const orc_sppm* r, const Isect* it, Sampler* sm, TraceCounters* tc) {
    const orc_scene* s = r->sc;
    if (s->n_lights == 0) return V(0, 0, 0);
    float uc = sampler_get1d(sm);
    float ulx, uly;
    sampler_get2d(sm, &ulx, &uly);
    float u0 = sampler_get1d(sm);
    return one_light_with(s, r->faithful, it, uc, ulx, uly, u0, tc);
}
void orc_sample_one_light(orc_scene* s, int64_t n, const float* org, const float* dir, const float* u, int32_t* prim,
                          float* ld, int faithful) {
    for (int64_t i = 0; i < n; ++i) {
        Ray r = make_ray(ld3(org + 3 * i), ld3(dir + 3 * i));
        Isect is; memset(&is, 0, sizeof(is)); is.prim = -1;
        int h = scene_intersect(s, &r, &is, faithful, 0);
        prim[i] = h ? is.prim : -1;
        v3 L = V(0, 0, 0);
        if (h && s->n_lights > 0) L = one_light_with(s, faithful, &is, u[4 * i], u[4 * i + 1], u[4 * i + 2], u[4 * i + 3], 0);
        st3(ld + 3 * i, L);
    }
}

/* One pixel of traceTile, src/tracer/TkSppmTracer.cpp:296-371.  With
 * queries != 0 the gathers are reported (for hitpoint comparison) instead of
 * executed and the pixel is left untouched except Ld (returned in ld_out). */
static int camera_path(const orc_sppm* r, int iter, int x, int y, TkSPPMPixel* p, CamThread* ct,
                       OrcQuery* queries, float* ld_out) {
    const orc_scene* s = r->sc;
    Sampler sm;
    sm.px = x; sm.py = y; sm.sample_idx = iter; sm.spp = r->num_recurse; sm.xs = r->xs; sm.ys = r->ys;
    sm.dim = 0; sm.seed = 0;
    sm.rng_seed = r->seed; sm.rng_pixel = (uint32_t)(x + y * s->w); sm.rng_iter = (uint32_t)iter; sm.rng_cost = r->rng_cost; sm.no_jitter = r->no_jitter;
    float cx, cy, lx, ly;
    sampler_get2d(&sm, &cx, &cy);
    cx = cx + (float)x; cy = cy + (float)y;
    sampler_get2d(&sm, &lx, &ly);          /* lens sample: consumed, unused with a pinhole */
    Ray ray = camera_ray(s, cx, cy);
    v3 coef = V(1.f, 1.f, 1.f);
    v3 Ld = V(0, 0, 0);
    int has_glossy = 1, nq = 0;
    for (int i = 0; i < r->max_depth; ++i) {
        Isect is; memset(&is, 0, sizeof(is));
        if (!scene_intersect(s, &ray, &is, r->faithful, &ct->tc)) break;
        const TkObject* ob = &s->objects[is.obj];
        const TkMaterial* m = &s->mats[ob->material];
        int diffuse = m->type == TK_DIFFUSE;
        v3 wo = vneg(ray.d);
        if (has_glossy && ob->light >= 0) Ld = vadd(Ld, vmul(coef, light_L(&s->lights[ob->light], is.n, wo)));
        if (has_glossy || !diffuse) Ld = vadd(Ld, vmul(coef, uniform_sample_one_light(r, &is, &sm, &ct->tc)));
        if (diffuse) {
            if (has_glossy && r->stored_c > 0) {
                if (queries) {
                    OrcQuery* q = &queries[nq++];
                    st3(q->p, is.p); st3(q->n, is.n); st3(q->wo, is.wo); st3(q->coef, coef);
                    memcpy(q->R, m->R, sizeof(float) * 3); q->pixel = x + y * s->w; q->which = 1;
                } else {
                    float radius2 = p->causticRadius2; int np;
                    v3 est = map_radiance_estimate(r->cmap, is.p, is.n, is.wo, ld3(m->R), &radius2, &np, ct->d2, ct->nb);
                    v3 lind = vmul(coef, est);
                    pixel_update_one(p->causticFlux, &p->causticPhotons, &p->causticRadius2, lind, np, radius2);
                    ct->gc++; ct->gphot += np;
                }
            }
            if (!has_glossy) {
                if (queries) {
                    OrcQuery* q = &queries[nq++];
                    st3(q->p, is.p); st3(q->n, is.n); st3(q->wo, is.wo); st3(q->coef, coef);
                    memcpy(q->R, m->R, sizeof(float) * 3); q->pixel = x + y * s->w; q->which = 0;
                } else {
                    float radius2 = p->globalRadius2; int np;
                    v3 est = map_radiance_estimate(r->gmap, is.p, is.n, is.wo, ld3(m->R), &radius2, &np, ct->d2, ct->nb);
                    v3 lind = vmul(coef, est);
                    pixel_update_one(p->globalFlux, &p->globalPhotons, &p->globalRadius2, lind, np, radius2);
                    ct->gg++; ct->gphot += np;
                }
                break;
            }
        }
        has_glossy &= !diffuse;
        v3 wi; float pdf;
        float ux, uy;
        sampler_get2d(&sm, &ux, &uy);
        v3 f = mat_sample_f(m, 0, wo, &wi, is.n, ux, uy, &pdf);
        if ((f.x == 0 && f.y == 0 && f.z == 0) || pdf == 0) break;
        coef = vdivs(vscale(vmul(coef, f), vabsdot(wi, is.n)), pdf);
        ray = spawn_ray(&is, wi);
    }
    if (ld_out) st3(ld_out, Ld);
    else { p->Ld[0] += Ld.x; p->Ld[1] += Ld.y; p->Ld[2] += Ld.z; }
    return nq;
}

typedef struct { orc_sppm* r; int iter; int y0; CamThread* cts; } CamJob;
static void camera_row_job(void* a, int64_t i, int tid) {
    CamJob* j = (CamJob*)a;
    orc_sppm* r = j->r;
    int y = j->y0 + (int)i;
    if (((y / 8) % r->world) != r->rank) return;
    for (int x = 0; x < r->sc->w; ++x)
        camera_path(r, j->iter, x, y, &r->pp[x + y * r->sc->w], &j->cts[tid], 0, 0);
}
void orc_sppm_camera_pass(orc_sppm* r, int iter, int y0, int y1, TkIterStats* st) {
    int T = r->threads < 1 ? 1 : (r->threads > 256 ? 256 : r->threads);
    int kmax = r->prm.globalSample > r->prm.causticSample ? r->prm.globalSample : r->prm.causticSample;
    CamThread* cts = (CamThread*)calloc((size_t)T, sizeof(CamThread));
    for (int t = 0; t < T; ++t) {
        cts[t].d2 = (float*)malloc(sizeof(float) * (size_t)(kmax + 1));
        cts[t].nb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(kmax + 1));
    }
    CamJob job = {r, iter, y0, cts};
    parallel_for(T, y1 - y0, 1, camera_row_job, &job);
    for (int t = 0; t < T; ++t) {
        r->tc_camera[0] += cts[t].tc.rays; r->tc_camera[1] += cts[t].tc.nodes; r->tc_camera[2] += cts[t].tc.prims;
        if (st) { st->rays += cts[t].tc.rays; st->rays_camera += cts[t].tc.rays; st->gathers_global += cts[t].gg; st->gathers_caustic += cts[t].gc; st->gather_photons += cts[t].gphot; }
        free(cts[t].d2); free(cts[t].nb);
    }
    free(cts);
}
int orc_sppm_camera_path(orc_sppm* r, int iter, int x, int y, float ld_out[3], OrcQuery q[2]) {
    CamThread ct; memset(&ct, 0, sizeof(ct));
    return camera_path(r, iter, x, y, 0, &ct, q, ld_out);
}

void orc_sppm_iteration(orc_sppm* r, int iter, TkIterStats* st) {
    orc_sppm_photon_pass(r, iter, st);
    orc_sppm_build_maps(r, st);
    orc_sppm_camera_pass(r, iter, 0, r->sc->h, st);
}

/* SppmTracer ctor + startRaytracing, src/tracer/TkSppmTracer.cpp:11-29, :385-398 */
orc_sppm* orc_sppm_begin(orc_scene* s, const TkSppmParam* param, int max_depth, float rr, uint64_t seed,
                         int rank, int world, int threads, int split_mode, int faithful_bvh, int rng_cost) {
    ensure_consts();
    orc_sppm* r = (orc_sppm*)calloc(1, sizeof(orc_sppm));
    r->sc = s; r->prm = *param; r->max_depth = max_depth; r->rr = rr; r->seed = (uint32_t)seed;
    r->rank = rank; r->world = world < 1 ? 1 : world; r->threads = threads;
    r->split_mode = split_mode; r->faithful = faithful_bvh; r->rng_cost = rng_cost;
    int x = 0, y = 0;
    while (++x * ++y < param->numRecurse) {}
    r->num_recurse = x * y; r->xs = x; r->ys = y;
    r->gmap = orc_map_create(param->globalSize, param->globalSample);
    r->cmap = orc_map_create(param->causticSize, param->causticSample);
    r->quota_g = param->globalSize / r->world + (rank < param->globalSize % r->world ? 1 : 0);
    r->quota_c = param->causticSize / r->world + (rank < param->causticSize % r->world ? 1 : 0);
    r->loc_g = (PhotonRec*)malloc(sizeof(PhotonRec) * (size_t)(r->quota_g ? r->quota_g : 1));
    r->loc_c = (PhotonRec*)malloc(sizeof(PhotonRec) * (size_t)(r->quota_c ? r->quota_c : 1));
    r->yield_g = 2.0; r->yield_c = 0.05;
    size_t np = (size_t)s->w * (size_t)s->h;
    r->pp = (TkSPPMPixel*)calloc(np ? np : 1, sizeof(TkSPPMPixel));
    for (size_t i = 0; i < np; ++i) { r->pp[i].causticRadius2 = param->radius2; r->pp[i].globalRadius2 = param->radius2; }
    return r;
}
/* jitter = 0: the camera sampler is the reference's StratifiedSampler built with jitter = false. */
/* rays, box tests and primitive tests of all photon passes (out[0..2]) and camera passes (out[3..5]) so far */
void orc_sppm_trace_counters(orc_sppm* r, int64_t out[6]) {
    for (int i = 0; i < 3; ++i) { out[i] = r->tc_photon[i]; out[3 + i] = r->tc_camera[i]; }
}
void orc_sppm_set_jitter(orc_sppm* r, int jitter) { r->no_jitter = jitter ? 0 : 1; }
void orc_sppm_set_tie_rule(orc_sppm* r, int by_index) { r->gmap->tie_by_index = by_index; r->cmap->tie_by_index = by_index; }
void orc_sppm_destroy(orc_sppm* r) {
    if (!r) return;
    orc_map_destroy(r->gmap); orc_map_destroy(r->cmap);
    free(r->loc_g); free(r->loc_c); free(r->pp); free(r);
}
void orc_sppm_read_pixels(orc_sppm* r, TkSPPMPixel* out) {
    memcpy(out, r->pp, sizeof(TkSPPMPixel) * (size_t)r->sc->w * (size_t)r->sc->h);
}
void orc_sppm_write_pixels(orc_sppm* r, const TkSPPMPixel* in) {
    memcpy(r->pp, in, sizeof(TkSPPMPixel) * (size_t)r->sc->w * (size_t)r->sc->h);
}
void orc_sppm_resolve(orc_sppm* r, uint8_t* out) {
    orc_resolve_rgb8(r->pp, r->sc->w, r->sc->h, r->num_recurse, r->shoot_g, r->shoot_c, (int)r->stored_c, out);
}
void orc_sppm_counters(orc_sppm* r, int64_t* sg, int64_t* sc, int32_t* nr) {
    if (sg) *sg = r->shoot_g;
    if (sc) *sc = r->shoot_c;
    if (nr) *nr = r->num_recurse;
}

/* ===================================================================== hash-grid SPPM
 * tk::SPPM (src/TkSPPM.h:10-64, src/TkSPPM.cpp:19-272, :347-465): visible point per pixel, spatial
 * hash grid over the visible points, photons deposited into the pixels they reach, per-pixel
 * radius reduction.  The class is not part of the reference's build (absent from
 * src/CMakeLists.txt, its line in src/TkLoader.cpp:250 commented out), but its translation unit compiles
 * with the rest: oracle/ref_probe2.cpp takes it in and runs its passes one by one with the unseeded
 * draws fed in, and tests/golden/ref_hsppm.npz pins this restatement to it (hash, worldToGrid, the pixel
 * state after the ray, photon and renew passes of two iterations).
 * An orc_sppm created with causticSize 0 carries the state; the TkSPPMPixel fields are reused:
 * Ld = Ld, causticFlux = Tau, causticPhotons = N, causticRadius2 = radius (not squared). */
typedef struct { v3 p, n, wo, thr; const TkMaterial* m; float radius; int pixel; } VisPoint;

static uint32_t hg_hash(int x, int y, int z, uint32_t buckets) {             /* src/TkSPPM.cpp:109-112 */
    int32_t h = (int32_t)(((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349663u) ^ ((uint32_t)z * 83492791u));
    return (uint32_t)((uint64_t)(int64_t)h % (uint64_t)buckets);             /* s32 widened to size_t */
}
typedef struct { box3 b; int res[3]; int valid; } HashGrid;
static int hg_world_to_grid(const HashGrid* g, v3 p, int pi[3]) {            /* src/TkSPPM.cpp:88-99 */
    float o[3] = {p.x - g->b.mn.x, p.y - g->b.mn.y, p.z - g->b.mn.z};        /* Bounds3::Offset */
    float mn[3] = {g->b.mn.x, g->b.mn.y, g->b.mn.z}, mx[3] = {g->b.mx.x, g->b.mx.y, g->b.mx.z};
    int in = 1;
    for (int i = 0; i < 3; ++i) {
        if (mx[i] > mn[i]) o[i] /= mx[i] - mn[i];
        pi[i] = (int)((float)g->res[i] * o[i]);
        in &= (pi[i] >= 0 && pi[i] < g->res[i]);
        pi[i] = pi[i] < 0 ? 0 : (pi[i] > g->res[i] - 1 ? g->res[i] - 1 : pi[i]);
    }
    return in;
}

/* SpatialHashGrids::worldToGrid + hash for n points in a grid of the given bounds (min xyz, max xyz) and
 * resolution, hashed into `buckets` lists: cell[3n], inside[n], hash[n].  For the parity tests. */
void orc_hg_cells(const float bounds[6], const int32_t res[3], uint32_t buckets, int64_t n, const float* points,
                  int32_t* cell, int32_t* inside, int32_t* hash) {
    HashGrid g;
    g.b.mn = V(bounds[0], bounds[1], bounds[2]); g.b.mx = V(bounds[3], bounds[4], bounds[5]);
    g.res[0] = res[0]; g.res[1] = res[1]; g.res[2] = res[2]; g.valid = 1;
    for (int64_t i = 0; i < n; ++i) {
        int pi[3];
        inside[i] = hg_world_to_grid(&g, V(points[3 * i], points[3 * i + 1], points[3 * i + 2]), pi);
        cell[3 * i] = pi[0]; cell[3 * i + 1] = pi[1]; cell[3 * i + 2] = pi[2];
        hash[i] = (int32_t)hg_hash(pi[0], pi[1], pi[2], buckets);
    }
}

/* One pixel of SPPM::traceTile, src/TkSPPM.cpp:140-177.  Returns 1 with *vp filled when the path
 * ends on a diffuse surface. */
static int hsppm_camera_path(const orc_sppm* r, int spp, int iter, int x, int y, TkSPPMPixel* p, VisPoint* vp, TraceCounters* tc) {
    const orc_scene* s = r->sc;
    Sampler sm;
    sm.px = x; sm.py = y; sm.sample_idx = iter; sm.spp = r->num_recurse; sm.xs = r->xs; sm.ys = r->ys;
    sm.dim = 0; sm.seed = 0;
    sm.rng_seed = r->seed; sm.rng_pixel = (uint32_t)(x + y * s->w); sm.rng_iter = (uint32_t)iter; sm.rng_cost = r->rng_cost; sm.no_jitter = r->no_jitter;
    float cx, cy, lx, ly;
    sampler_get2d(&sm, &cx, &cy);
    cx = cx + (float)x; cy = cy + (float)y;
    sampler_get2d(&sm, &lx, &ly);
    Ray ray = camera_ray(s, cx, cy);
    v3 coef = V(1.f, 1.f, 1.f);
    int has_specular = 1, found = 0;
    for (int i = 0; i < r->max_depth; ++i) {
        Isect is; memset(&is, 0, sizeof(is));
        if (!scene_intersect(s, &ray, &is, r->faithful, tc)) break;
        const TkObject* ob = &s->objects[is.obj];
        const TkMaterial* m = &s->mats[ob->material];
        int diffuse = m->type == TK_DIFFUSE;
        v3 wo = vneg(ray.d);
        v3 Ld = V(0, 0, 0);
        if (has_specular && ob->light >= 0) Ld = vadd(Ld, vmul(coef, light_L(&s->lights[ob->light], is.n, wo)));
        Ld = vadd(Ld, vmul(coef, uniform_sample_one_light(r, &is, &sm, tc)));
        Ld = vdivs(Ld, (float)spp);                       /* spp is a power of two: same as dividing each term */
        p->Ld[0] += Ld.x; p->Ld[1] += Ld.y; p->Ld[2] += Ld.z;
        if (diffuse) {
            vp->p = is.p; vp->n = is.n; vp->wo = is.wo; vp->thr = coef; vp->m = m;
            vp->radius = p->causticRadius2; vp->pixel = x + y * s->w;
            found = 1;
            break;
        }
        has_specular &= (m->type & (TK_MIRROR | TK_GLASS)) != 0;
        v3 wi; float pdf;
        float ux, uy;
        sampler_get2d(&sm, &ux, &uy);
        v3 f = mat_sample_f(m, 0, wo, &wi, is.n, ux, uy, &pdf);
        if ((f.x == 0 && f.y == 0 && f.z == 0) || pdf == 0) break;
        coef = vdivs(vscale(vmul(coef, f), vabsdot(wi, is.n)), pdf);
        ray = spawn_ray(&is, wi);
    }
    return found;
}

typedef struct { orc_sppm* r; int spp, iter; VisPoint* vis; uint8_t* has; TraceCounters* tcs; } HCamJob;
static void hsppm_row_job(void* a, int64_t y, int tid) {
    HCamJob* j = (HCamJob*)a;
    orc_sppm* r = j->r;
    int w = r->sc->w;
    for (int x = 0; x < w; ++x) {
        int px = x + (int)y * w;
        j->has[px] = (uint8_t)hsppm_camera_path(r, j->spp, j->iter, x, (int)y, &r->pp[px], &j->vis[px], &j->tcs[tid]);
    }
}

/* One iteration of SPPM::render, src/TkSPPM.cpp:347-431: ray pass, grid rebuild, photon pass, renew
 * pass.  Photons are traced one after the other in index order (the reference's threads interleave
 * them under a mutex), the Russian roulette draws are hashed as everywhere in this file.
 * stats[0] = visible points, [1] = grid entries, [2] = photon deposits (sum of M). */
void orc_hsppm_iteration(orc_sppm* r, int spp, int iter, int64_t stats[3]) {
    const orc_scene* s = r->sc;
    const int np = s->w * s->h;
    int T = r->threads < 1 ? 1 : (r->threads > 256 ? 256 : r->threads);
    VisPoint* vis = (VisPoint*)calloc((size_t)np, sizeof(VisPoint));
    uint8_t* has = (uint8_t*)calloc((size_t)np, 1);
    TraceCounters* tcs = (TraceCounters*)calloc((size_t)T, sizeof(TraceCounters));
    HCamJob job = {r, spp, iter, vis, has, tcs};
    parallel_for(T, s->h, 1, hsppm_row_job, &job);
    free(tcs);
    /* rebuild, src/TkSPPM.cpp:45-86 */
    HashGrid g; g.b = box_empty(); g.valid = 0;
    float max_radius = -INFINITY;
    int64_t nvis = 0;
    for (int i = 0; i < np; ++i) if (has[i]) {
        v3 rr3 = V(vis[i].radius, vis[i].radius, vis[i].radius);
        g.b = box_union_p(g.b, vsub(vis[i].p, rr3));
        g.b = box_union_p(g.b, vadd(vis[i].p, rr3));
        max_radius = tk_max(max_radius, vis[i].radius);
        ++nvis;
    }
    const uint32_t buckets = (uint32_t)np;
    int32_t* start = (int32_t*)calloc((size_t)buckets + 1, sizeof(int32_t));
    int32_t* fill = (int32_t*)calloc((size_t)buckets + 1, sizeof(int32_t));
    int32_t* items = 0;
    int64_t entries = 0;
    if (nvis > 0) {
        g.valid = 1;
        v3 diag = vsub(g.b.mx, g.b.mn);
        int ext = box_max_extent(g.b);
        float max_diag = vget(diag, ext);
        int base = (int)(max_diag / max_radius);
        for (int a = 0; a < 3; ++a) {
            int rs = (int)((float)base * vget(diag, a) / max_diag);
            g.res[a] = rs > 1 ? rs : 1;
        }
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = 0; i < np; ++i) if (has[i]) {
                v3 rr3 = V(vis[i].radius, vis[i].radius, vis[i].radius);
                int lo[3], hi[3];
                hg_world_to_grid(&g, vsub(vis[i].p, rr3), lo);
                hg_world_to_grid(&g, vadd(vis[i].p, rr3), hi);
                for (int z = lo[2]; z <= hi[2]; ++z)
                    for (int y = lo[1]; y <= hi[1]; ++y)
                        for (int x = lo[0]; x <= hi[0]; ++x) {
                            uint32_t b = hg_hash(x, y, z, buckets);
                            if (pass == 0) start[b + 1]++;
                            else items[start[b] + fill[b]++] = i;
                        }
            }
            if (pass == 0) {
                for (uint32_t b = 0; b < buckets; ++b) start[b + 1] += start[b];
                entries = start[buckets];
                items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(entries ? entries : 1));
            }
        }
    }
    /* photon pass, src/TkSPPM.cpp:181-250 */
    float* phi = (float*)calloc((size_t)np * 3, sizeof(float));
    int32_t* M = (int32_t*)calloc((size_t)np, sizeof(int32_t));
    const int64_t per_it = r->prm.globalSize;
    int64_t deposits = 0;
    TraceCounters tc; memset(&tc, 0, sizeof(tc));
    const int nl = s->n_lights;
    for (int64_t k = 0; k < per_it && nl > 0; ++k) {
        uint64_t H = (uint64_t)iter * (uint64_t)per_it + (uint64_t)k;
        int dim = 0;
        int li = (int)(orc_radical_inverse(dim++, H) * nl);
        if (li > nl - 1) li = nl - 1;
        float pdf = 1.0f / nl;
        float u0x = orc_radical_inverse(dim, H), u0y = orc_radical_inverse(dim + 1, H);
        float u1x = orc_radical_inverse(dim + 2, H), u1y = orc_radical_inverse(dim + 3, H);
        dim += 4;
        Ray ray; v3 normal; float pos_pdf, dir_pdf;
        float upick = photon_rng(r, H, 2, 0, 0);
        v3 weight = light_sample_le(s, li, u0x, u0y, u1x, u1y, upick, &ray, &normal, &pos_pdf, &dir_pdf);
        if (pos_pdf == 0 || dir_pdf == 0) continue;   /* the reference returns from its whole batch here; a light with zero pdf does not occur */
        weight = vdivs(vscale(weight, vabsdot(ray.d, normal)), pos_pdf * dir_pdf * pdf);
        for (int i = 0; i < r->max_depth; ++i) {
            Isect is; memset(&is, 0, sizeof(is));
            if (!scene_intersect(s, &ray, &is, r->faithful, &tc)) break;
            v3 wo = vneg(ray.d);
            if (i > 0 && g.valid) {
                int pi[3];
                if (hg_world_to_grid(&g, is.p, pi)) {
                    uint32_t b = hg_hash(pi[0], pi[1], pi[2], buckets);
                    for (int32_t e = start[b]; e < start[b + 1]; ++e) {
                        VisPoint* v = &vis[items[e]];
                        v3 d = vsub(v->p, is.p);
                        if (vdot(d, d) <= v->radius * v->radius) {
                            M[items[e]]++; ++deposits;
                            v3 f = mat_f(v->m, v->wo, wo, v->n);
                            phi[3 * items[e]] += f.x * weight.x;
                            phi[3 * items[e] + 1] += f.y * weight.y;
                            phi[3 * items[e] + 2] += f.z * weight.z;
                        }
                    }
                }
            }
            if (photon_rng(r, H, 2, i, 1) > r->rr) break;
            v3 wi; float bpdf = 0;
            float ux = orc_radical_inverse(dim, H), uy = orc_radical_inverse(dim + 1, H);   /* the same two dimensions at every bounce, as there */
            const TkMaterial* m = &s->mats[s->objects[is.obj].material];
            v3 f = mat_sample_f(m, 1, wo, &wi, is.n, ux, uy, &bpdf);
            if (bpdf == 0 || (f.x == 0 && f.y == 0 && f.z == 0)) break;
            weight = vdivs(vscale(vmul(weight, f), vabsdot(wi, is.n)), bpdf * r->rr);
            ray = spawn_ray(&is, wi);
        }
    }
    r->shoot_g += per_it;
    r->tc_photon[0] += tc.rays; r->tc_photon[1] += tc.nodes; r->tc_photon[2] += tc.prims;
    /* renew pass, src/TkSPPM.cpp:252-272 */
    for (int i = 0; i < np; ++i) {
        if (!has[i] || M[i] <= 0) continue;
        TkSPPMPixel* p = &r->pp[i];
        const float alpha = 0.7f;
        float N = p->causticPhotons, radius = p->causticRadius2;
        float newN = N + alpha * (float)M[i];
        float newR = radius * sqrtf(newN / (N + (float)M[i]));
        float thr[3] = {vis[i].thr.x, vis[i].thr.y, vis[i].thr.z};
        for (int c = 0; c < 3; ++c)
            p->causticFlux[c] = (p->causticFlux[c] + thr[c] * phi[3 * i + c]) * (newR * newR) / (radius * radius);
        p->causticPhotons = newN;
        p->causticRadius2 = newR;
    }
    if (stats) { stats[0] = nvis; stats[1] = entries; stats[2] = deposits; }
    free(vis); free(has); free(start); free(fill); free(items); free(phi); free(M);
}

/* The final image of SPPM::render, src/TkSPPM.cpp:436-450 (rows bottom-up). */
void orc_hsppm_resolve(orc_sppm* r, uint8_t* out) {
    const orc_scene* s = r->sc;
    int off = 0;
    for (int y = s->h; y-- > 0;)
        for (int x = 0; x < s->w; ++x) {
            const TkSPPMPixel* p = &r->pp[x + y * s->w];
            float radius = p->causticRadius2;
            float denom = (float)(r->num_recurse * r->prm.globalSize) * k_pi * radius * radius;
            for (int c = 0; c < 3; ++c) {
                float v = p->Ld[c] / (float)r->num_recurse + p->causticFlux[c] / denom;
                float cl = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
                out[off++] = (uint8_t)(255 * powf(cl, 0.6f));
            }
        }
}

/* ===================================================================== path tracer
 * PathTracer (src/tracer/TkPathTracer.cpp:13-24, :127-204) with the film of the headless program:
 * BoxFilter(0.5, 0.5) (src/main.cpp:41), FilmTile::addSample / Film::mergeFilmTile / Film::writeImage
 * (src/core/TkFilm.cpp:14-42, :108-153), FilterTable (src/core/TkFilter.cpp:13-32).  The Russian
 * roulette draw (get_random_float, unseeded in the reference) is the hashed number the GPU uses.
 * Samples are added to the film pixel by pixel in sample order (the reference adds them tile by
 * tile; the sums differ in the last bits only when a sample lands on a pixel edge).
 * Uses the scene, depth, roulette probability and seed of an orc_sppm. */
static v3 pt_li(const orc_sppm* r, Ray ray, Sampler* sm, int pixel, int sample, TraceCounters* tc) {
    const orc_scene* s = r->sc;
    v3 L = V(0, 0, 0), beta = V(1, 1, 1);
    int test = 1;
    for (int bounces = 0; bounces < r->max_depth; ++bounces) {
        Isect is; memset(&is, 0, sizeof(is));
        if (!scene_intersect(s, &ray, &is, r->faithful, tc)) break;
        v3 wo = vneg(ray.d);
        const TkObject* ob = &s->objects[is.obj];
        const TkMaterial* m = &s->mats[ob->material];
        /* the GPU adds the two terms of a hit as one sum to the film; same grouping here */
        v3 here = V(0, 0, 0);
        if (test && ob->light >= 0) here = vadd(here, vmul(beta, light_L(&s->lights[ob->light], is.n, wo)));
        here = vadd(here, vmul(beta, uniform_sample_one_light(r, &is, sm, tc)));
        L = vadd(L, here);
        if (orc_rng_u01(r->seed, (uint32_t)pixel, (uint32_t)sample, 0x10000u + (uint32_t)bounces) > r->rr) break;
        v3 wi; float pdf;
        float ux, uy;
        sampler_get2d(sm, &ux, &uy);
        v3 f = mat_sample_f(m, 0, wo, &wi, is.n, ux, uy, &pdf);
        if (pdf == 0 || (f.x == 0 && f.y == 0 && f.z == 0)) break;
        test = (m->type & (TK_GLASS | TK_MIRROR)) != 0;
        beta = vmul(beta, vdivs(vscale(f, vabsdot(wi, is.n)), pdf * r->rr));
        ray = spawn_ray(&is, wi);
    }
    return L;
}

typedef struct { orc_sppm* r; int spp, xs, ys, s0, s1; float* contrib; float* wsum; pthread_mutex_t* mu; TraceCounters* tcs; } PtJob;
static void pt_row_job(void* a, int64_t yy, int tid) {
    PtJob* j = (PtJob*)a;
    orc_sppm* r = j->r;
    const orc_scene* s = r->sc;
    const int W = s->w, H = s->h, y = (int)yy;
    /* FilterTable of the box filter: every entry 1 / normalizeTerm = 1 / (4 * 0.5 * 0.5) */
    const float weight = 1.0f * (1.0f / (4.0f * 0.5f * 0.5f));
    for (int x = 0; x < W; ++x) {
        for (int k = j->s0; k < j->s1; ++k) {
            Sampler sm;
            sm.px = x; sm.py = y; sm.sample_idx = k; sm.spp = j->spp; sm.xs = j->xs; sm.ys = j->ys; sm.dim = 0; sm.seed = 0;
            sm.rng_seed = r->seed; sm.rng_pixel = (uint32_t)(x + y * W); sm.rng_iter = (uint32_t)k; sm.rng_cost = r->rng_cost; sm.no_jitter = r->no_jitter;
            float cx, cy, lx, ly;
            sampler_get2d(&sm, &cx, &cy);
            cx = cx + (float)x; cy = cy + (float)y;
            sampler_get2d(&sm, &lx, &ly);
            Ray ray = camera_ray(s, cx, cy);
            v3 L = pt_li(r, ray, &sm, x + y * W, k, &j->tcs[tid]);
            /* FilmTile::addSample */
            float dx = cx - 0.5f, dy = cy - 0.5f;
            int p0x = (int)ceilf(dx - 0.5f), p0y = (int)ceilf(dy - 0.5f);
            int p1x = (int)floorf(dx + 0.5f + 1), p1y = (int)floorf(dy + 0.5f + 1);
            if (p0x < 0) p0x = 0;
            if (p0y < 0) p0y = 0;
            if (p1x > W) p1x = W;
            if (p1y > H) p1y = H;
            int own = p0x == x && p1x == x + 1 && p0y == y && p1y == y + 1;
            if (!own) pthread_mutex_lock(j->mu);   /* a neighbour's row may be another thread's */
            for (int py = p0y; py < p1y; ++py)
                for (int px = p0x; px < p1x; ++px) {
                    float* c = &j->contrib[3 * (px + py * W)];
                    c[0] += L.x * weight; c[1] += L.y * weight; c[2] += L.z * weight;
                    j->wsum[px + py * W] += weight;
                }
            if (!own) pthread_mutex_unlock(j->mu);
        }
    }
}
/* Samples [s0, s1) of every pixel into the film (contrib: 3 floats per pixel, wsum: 1). */
void orc_pt_render(orc_sppm* r, int spp, int s0, int s1, float* contrib, float* wsum) {
    int T = r->threads < 1 ? 1 : (r->threads > 256 ? 256 : r->threads);
    int x = 1, y = spp / x;                                  /* src/tracer/TkPathTracer.cpp:16-22 */
    while (y != x && y / x != 2) { x <<= 1; y = spp / x; }
    pthread_mutex_t mu;
    pthread_mutex_init(&mu, 0);
    TraceCounters* tcs = (TraceCounters*)calloc((size_t)T, sizeof(TraceCounters));
    PtJob job = {r, spp, x, y, s0, s1, contrib, wsum, &mu, tcs};
    /* rows that share an edge sample are serialised by the mutex; a row's own sums are only ever
     * touched by the thread that has the row, except for those edge samples */
    parallel_for(T, r->sc->h, 1, pt_row_job, &job);
    for (int t = 0; t < T; ++t) { r->tc_camera[0] += tcs[t].rays; r->tc_camera[1] += tcs[t].nodes; r->tc_camera[2] += tcs[t].prims; }
    free(tcs);
    pthread_mutex_destroy(&mu);
}
/* Film::writeImage, src/core/TkFilm.cpp:124-153 (square images: its x loop runs over the y bounds). */
void orc_pt_resolve(const float* contrib, const float* wsum, int w, int h, uint8_t* out) {
    int off = 0;
    for (int y = h; y-- > 0;)
        for (int x = 0; x < w; ++x) {
            int idx = x + y * w;
            for (int c = 0; c < 3; ++c) {
                float v = contrib[3 * idx + c];
                if (wsum[idx] != 0) v = v / wsum[idx];
                float cl = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
                out[off++] = (uint8_t)(255 * powf(cl, 0.6f));
            }
        }
}
