// tk_device.h — device-side math of the SPPM path: vectors, the counter-based
// RNG, Halton / stratified sampling, materials, light sampling and ray
// offsetting.  Everything here is compiled with -ffp-contract=off so that a
// product followed by a sum rounds twice, like the CPU renderer's baseline
// x86-64 code; fused operations are spelled fmaf() where they are wanted.
// Citations are to the upstream CPU renderer (src/...).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tk_common.h"

namespace tkd {

struct F3 {
    float x, y, z;
};

#define TKD __device__ __forceinline__

constexpr float kPi = 3.14159274101257324f;        // float(4*atan(1)), src/TkMath.cpp:7
constexpr float kInvPi = 1.0f / kPi;               // src/TkMath.cpp:8
constexpr float kTwoPi = 6.28318548202514648f;     // float(2.0*pi)
constexpr float kHalfPi = 1.57079637050628662f;    // float(0.5*pi)
constexpr float kEps = 5.9604644775390625e-8f;     // epsilon*0.5, src/TkMath.cpp:14
constexpr float kOneMinusEps = 0.99999994039535522f;  // src/TkMath.cpp:15

TKD F3 mk3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
TKD F3 add3(F3 a, F3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
TKD F3 sub3(F3 a, F3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
TKD F3 mul3(F3 a, F3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
TKD F3 scale3(F3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
TKD F3 divs3(F3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
TKD F3 neg3(F3 a) { return mk3(-a.x, -a.y, -a.z); }
TKD F3 abs3(F3 a) { return mk3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
TKD float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }        // src/Vector.hpp:91-94
TKD float absdot3(F3 a, F3 b) { return fabsf(a.x * b.x + a.y * b.y + a.z * b.z); }
TKD F3 cross3(F3 a, F3 b) {                                                      // src/Vector.hpp:96-103
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
TKD F3 normalize3(F3 v) {                                                        // src/Vector.hpp:80-89
    float mag2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (mag2 > 0) {
        float inv = 1 / sqrtf(mag2);
        return mk3(v.x * inv, v.y * inv, v.z * inv);
    }
    return v;
}
TKD float norm3(F3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
TKD bool isblack(F3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
TKD F3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
TKD float comp3(F3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

TKD float sse_min(float a, float b) { return a < b ? a : b; }   // src/TkMath.h:11-16
TKD float sse_max(float a, float b) { return a > b ? a : b; }   // src/TkMath.h:18-23
TKD float gamma_n(int n) { return (n * kEps) / (1 - n * kEps); }   // src/TkMath.h:201-204

TKD float next_up(float v) {                                   // src/TkMath.h:334-346
    if (isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = __float_as_uint(v);
    if (v >= 0) ++ui; else --ui;
    return __uint_as_float(ui);
}
TKD float next_down(float v) {                                 // src/TkMath.h:348-358
    if (isinf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = __float_as_uint(v);
    if (v > 0) --ui; else ++ui;
    return __uint_as_float(ui);
}

TKD void coordinate_system(F3 v1, F3* v2, F3* v3) {            // src/Vector.hpp:116-123
    float sign = copysignf(1.0f, v1.z);
    float a = -1 / (sign + v1.z);
    float b = v1.x * v1.y * a;
    *v2 = mk3(1 + sign * v1.x * v1.x * a, sign * b, -sign * v1.x);
    *v3 = mk3(b, sign + v1.y * v1.y * a, -v1.y);
}

// ------------------------------------------------------------------- rng
// Counter-based stand-in for get_random_float() (src/TkMath.h:312-319, which
// seeds a new Mersenne Twister from the OS per call and so has no stream to
// reproduce): 24 uniform bits from three rounds of a 32-bit finaliser.
TKD uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
TKD float rng_u01(uint32_t seed, uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = seed ^ 0x9E3779B9u;
    h = mix32(h + a * 0x85EBCA6Bu);
    h = mix32(h + b * 0xC2B2AE35u);
    h = mix32(h + c * 0x27D4EB2Fu);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// Random numbers of light path H: slot 0 = light-triangle pick
// (src/Triangle.cpp:192), slot 1 = Russian roulette of bounce `bounce`
// (src/tracer/TkSppmTracer.cpp:144, :204).
TKD float photon_rng(uint32_t seed, uint64_t H, int kind, int bounce, int slot) {
    uint32_t b = (uint32_t)kind | ((uint32_t)bounce << 4) | ((uint32_t)(H >> 32) << 16);
    return rng_u01(seed, (uint32_t)H, b, (uint32_t)slot);
}

// ------------------------------------------------------- low discrepancy
template <int BASE>
TKD float radinv_base(uint64_t a) {                            // src/core/TkLowdiscrepancy.h:19-32
    const float inv_base = (float)1 / (float)BASE;
    float inv_base_n = 1;
    uint64_t rev = 0;
    if ((a >> 32) == 0) {   // same arithmetic in 32-bit digits while the index fits
        uint32_t a32 = (uint32_t)a;
        while (a32) {
            uint32_t next = a32 / (uint32_t)BASE;
            uint32_t digit = a32 - next * (uint32_t)BASE;
            rev = rev * (uint64_t)BASE + digit;
            inv_base_n *= inv_base;
            a32 = next;
        }
    } else {
        while (a) {
            uint64_t next = a / (uint64_t)BASE;
            uint64_t digit = a - next * (uint64_t)BASE;
            rev = rev * (uint64_t)BASE + digit;
            inv_base_n *= inv_base;
            a = next;
        }
    }
    return sse_min((float)rev * inv_base_n, kOneMinusEps);
}
// Base 2: the digits in reverse are the bits in reverse, and the running product of 1/2 is a power of two:
// the same integer and the same factor as the loop above makes, without its thirty turns.
TKD float radinv_base2(uint64_t a) {
    if (a == 0) return 0.f;
    const int n = 64 - __clzll((long long)a);                   // number of digits
    const uint64_t rev = __brevll(a) >> (64 - n);
    const float inv_base_n = __uint_as_float((uint32_t)(127 - n) << 23);   // 2^-n, n <= 64
    return sse_min((float)rev * inv_base_n, kOneMinusEps);
}
TKD float radical_inverse(int dim, uint64_t a) {
    switch (dim) {
        case 0: return radinv_base2(a);
        case 1: return radinv_base<3>(a);
        case 2: return radinv_base<5>(a);
        case 3: return radinv_base<7>(a);
        case 4: return radinv_base<11>(a);
        case 5: return radinv_base<13>(a);
        case 6: return radinv_base<17>(a);
        case 7: return radinv_base<19>(a);
        case 8: return radinv_base<23>(a);
        case 9: return radinv_base<29>(a);
        case 10: return radinv_base<31>(a);
        case 11: return radinv_base<37>(a);
        case 12: return radinv_base<41>(a);
        case 13: return radinv_base<43>(a);
        case 14: return radinv_base<47>(a);
        default: return radinv_base<53>(a);
    }
}

// Hash(currentPixel, dim, seed): MurmurHash64A over 16 bytes
// (src/TkHash.h:10-63, :111-119; src/core/sampler.cpp:48).
TKD uint64_t hash_pixel(int x, int y, int dim, int seed) {
    const uint64_t m = 0xc6a4a7935bd1e995ull;
    const int r = 47;
    uint64_t h = 0 ^ (16 * m);
    uint64_t k = (uint64_t)(uint32_t)x | ((uint64_t)(uint32_t)y << 32);
    k *= m; k ^= k >> r; k *= m; h ^= k; h *= m;
    k = (uint64_t)(uint32_t)dim | ((uint64_t)(uint32_t)seed << 32);
    k *= m; k ^= k >> r; k *= m; h ^= k; h *= m;
    h ^= h >> r; h *= m; h ^= h >> r;
    return h;
}
TKD int permutation_element(uint32_t i, uint32_t l, uint32_t p) {   // src/TkMath.h:206-234
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893du; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8;
        i *= 0x0929eb3fu; i ^= p >> 23; i ^= (i & w) >> 1; i *= 1 | p >> 27;
        i *= 0x6935fa69u; i ^= (i & w) >> 11; i *= 0x74dcb303u; i ^= (i & w) >> 2;
        i *= 0x9e501cc3u; i ^= (i & w) >> 2; i *= 0xc860a3dfu; i &= w; i ^= i >> 5;
    } while (i >= l);
    // (i + p) % l, without the division when l is a power of two (then l - 1 == w)
    return (int)((l & w) == 0 ? ((i + p) & w) : ((i + p) % l));
}

// StratifiedSampler of the thread that calls render() (seed 0), jittered
// (src/core/sampler.cpp:40-64).  Jitter slot = 2*dim (+1 for the second axis).
struct Sampler {
    int px, py, sample_idx, spp, xs, ys, dim;
    uint32_t rng_seed, rng_pixel, rng_iter;
    bool no_jitter;   // StratifiedSampler(..., jitter = false): stratum centres (src/core/sampler.cpp:51, :61-62)
};
TKD float sampler_jitter(const Sampler& s, int slot) {
    return s.no_jitter ? 0.5f : rng_u01(s.rng_seed, s.rng_pixel, s.rng_iter, (uint32_t)slot);
}
TKD float sampler_get1d(Sampler& s) {
    uint64_t hash = hash_pixel(s.px, s.py, s.dim, 0);
    int stratum = permutation_element((uint32_t)s.sample_idx, (uint32_t)s.spp, (uint32_t)hash);
    int d = s.dim;
    ++s.dim;
    float delta = sampler_jitter(s, 2 * d);
    return (stratum + delta) / s.spp;
}
TKD void sampler_get2d(Sampler& s, float* ox, float* oy) {
    uint64_t hash = hash_pixel(s.px, s.py, s.dim, 0);
    int stratum = permutation_element((uint32_t)s.sample_idx, (uint32_t)s.spp, (uint32_t)hash);
    int d = s.dim;
    s.dim += 2;
    int x, y;
    if ((s.xs & (s.xs - 1)) == 0) { x = stratum & (s.xs - 1); y = stratum >> (__ffs(s.xs) - 1); }   // no division for 2^n strata
    else { x = stratum % s.xs; y = stratum / s.xs; }
    float dx = sampler_jitter(s, 2 * d);
    float dy = sampler_jitter(s, 2 * d + 1);
    *ox = (x + dx) / s.xs;
    *oy = (y + dy) / s.ys;
}

// ------------------------------------------------------------- sampling
// Correctly rounded float sine / cosine / arc functions: the double function,
// rounded to float.  The CPU code's unqualified cos / sin / acos / atan2 on
// float arguments are the float overloads of the host's libm, whose last bit
// differs from platform to platform; this form gives the same value on the
// host and on the GPU.
TKD float tk_cosf(float x) { return (float)cos((double)x); }
TKD float tk_sinf(float x) { return (float)sin((double)x); }
TKD float tk_acosf(float x) { return (float)acos((double)x); }
TKD float tk_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
TKD void concentric_disk(float ux, float uy, float* dx, float* dy) {   // src/core/sampling.cpp:60-77
    float ox = ux * 2.f - 1.0f, oy = uy * 2.f - 1.0f;
    if (ox == 0.0f && oy == 0.f) { *dx = ox; *dy = oy; return; }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) {
        r = ox;
        theta = (float)((double)kHalfPi * 0.5 * (double)(oy / ox));
    } else {
        r = oy;
        theta = (float)((double)kHalfPi - (double)kHalfPi * 0.5 * (double)(ox / oy));
    }
    *dx = r * tk_cosf(theta);
    *dy = r * tk_sinf(theta);
}
TKD F3 cosine_hemisphere(float ux, float uy) {                 // src/core/sampling.h:32-37
    float dx, dy;
    concentric_disk(ux, uy, &dx, &dy);
    float z = sqrtf(sse_max(0.f, 1 - dx * dx - dy * dy));
    return mk3(dx, dy, z);
}

// ------------------------------------------------------------ materials
TKD F3 mat_reflect(F3 wo, F3 N) { return sub3(scale3(N, 2 * dot3(wo, N)), wo); }   // src/core/Material.hpp:77-80
TKD bool mat_refract(F3 wi, F3 n, float eta, F3* wt) {         // src/core/Material.hpp:82-92
    float cos_i = dot3(wi, n);
    float sin2_i = sse_max(0.f, 1 - cos_i * cos_i);
    float sin2_t = eta * eta * sin2_i;
    if (sin2_t >= 1) return false;
    float cos_t = sqrtf(1 - sin2_t);
    *wt = add3(scale3(neg3(wi), eta), scale3(n, eta * cos_i - cos_t));
    return true;
}
TKD F3 to_world(F3 a, F3 N) {                                  // src/core/Material.hpp:94-99
    F3 B, C;
    coordinate_system(N, &B, &C);
    return add3(add3(scale3(B, a.x), scale3(C, a.y)), scale3(N, a.z));
}
TKD float fr_dielectric(float cos_i, float eta_i, float eta_t) {   // src/core/Material.cpp:41-61
    cos_i = sse_max(sse_min(cos_i, 1.0f), -1.0f);
    bool entering = cos_i > 0.f;
    if (!entering) { float t = eta_i; eta_i = eta_t; eta_t = t; cos_i = fabsf(cos_i); }
    float sin_i = sqrtf(sse_max(0.0f, 1 - cos_i * cos_i));
    float sin_t = eta_i / eta_t * sin_i;
    if (sin_t >= 1) return 1;
    float cos_t = sqrtf(sse_max(0.0f, 1 - sin_t * sin_t));
    float rparl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    float rperp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (rparl * rparl + rperp * rperp) / 2;
}
// DiffuseMaterial::f (src/core/Material.cpp:36-39); specular materials return black.
TKD F3 mat_f(const TkMaterial& m, F3 wo, F3 wi, F3 N) {
    if (m.type != TK_DIFFUSE) return mk3(0, 0, 0);
    return dot3(wo, N) * dot3(wi, N) > 0 ? scale3(ld3(m.R), kInvPi) : mk3(0, 0, 0);
}
// sample_f: diffuse src/core/Material.cpp:22-34, glass :63-88, mirror :90-95.
// importance = true is the photon-tracing transport mode.
TKD F3 mat_sample_f(const TkMaterial& m, bool importance, F3 wo, F3* wi, F3 N, float ux, float uy, float* pdf) {
    *pdf = 0;
    *wi = mk3(0, 0, 0);
    if (m.type == TK_DIFFUSE) {
        F3 tmp = cosine_hemisphere(ux, uy);
        if (tmp.z < 0) tmp.z *= -1;
        *wi = to_world(tmp, N);
        *pdf = dot3(wo, N) * dot3(*wi, N) > 0 ? dot3(*wi, N) * kInvPi : 0;
        return mat_f(m, wo, *wi, N);
    } else if (m.type == TK_GLASS) {
        float pr = fr_dielectric(dot3(wo, N), m.etaA, m.etaB);
        if (ux < pr) {
            *wi = mat_reflect(wo, N);
            *pdf = pr;
            return divs3(scale3(ld3(m.R), pr), absdot3(*wi, N));
        }
        bool entering = dot3(wo, N) > 0;
        float etai = entering ? m.etaA : m.etaB;
        float etat = entering ? m.etaB : m.etaA;
        if (!mat_refract(wo, entering ? N : neg3(N), etai / etat, wi)) return mk3(0, 0, 0);
        F3 ft = scale3(ld3(m.T), 1 - pr);
        if (!importance) ft = scale3(ft, (etai * etai) / (etat * etat));
        *pdf = 1 - pr;
        return divs3(ft, absdot3(*wi, N));
    }
    *wi = mat_reflect(wo, N);
    *pdf = 1;
    return divs3(ld3(m.R), absdot3(*wi, N));
}

// -------------------------------------------------------------- ray spawn
TKD F3 offset_ray_origin(F3 p, F3 perr, F3 n, F3 w) {          // src/Ray.hpp:43-57
    float d = dot3(abs3(n), perr);
    F3 off = scale3(n, d);
    if (dot3(w, n) < 0) off = neg3(off);
    F3 po = add3(p, off);
    if (off.x > 0) po.x = next_up(po.x); else if (off.x < 0) po.x = next_down(po.x);
    if (off.y > 0) po.y = next_up(po.y); else if (off.y < 0) po.y = next_down(po.y);
    if (off.z > 0) po.z = next_up(po.z); else if (off.z < 0) po.z = next_down(po.z);
    return po;
}

// Photon direction quantisation (src/tracer/TkSppmTracer.cpp:135-140): float
// acos / atan2 and float products; acos of a value just above one gives NaN,
// which the CPU code converts with undefined behaviour: clamp.
TKD uint32_t encode_direction(F3 wo) {
    float th = 255 * tk_acosf(wo.z) * kInvPi;
    int t = (th != th) ? 255 : (int)th;
    t = t > 255 ? 255 : (t < 0 ? 0 : t);
    int p = (int)(255 * tk_atan2f(wo.y, wo.x) / (2 * kPi));
    if (p < 0) p += 255;
    if (p > 255) p = 255;
    return (uint32_t)t | ((uint32_t)p << 8);
}

// Matrix4::concatenatePos / transformDirectionAffine (src/Matrix4.cpp:118-128,
// src/Matrix4.h:247-253)
TKD F3 mat_pos(const float* m, F3 p) {
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1) return mk3(xp, yp, zp);
    return divs3(mk3(xp, yp, zp), wp);
}
TKD F3 mat_dir(const float* m, F3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
               m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// Camera::generateRay with a pinhole (src/core/camera.cpp:51-71, src/main.cpp:46)
TKD void camera_ray(const DCamera& cam, float fx, float fy, F3* o, F3* d) {
    F3 dc = normalize3(mat_pos(cam.r2c, mk3(fx, fy, 0)));
    *o = mat_pos(cam.c2w, mk3(0, 0, 0));
    *d = normalize3(mat_dir(cam.c2w, dc));
}

// Wave-aggregated append: one atomic per wave, returns this lane's slot.
TKD int wave_append(int* counter, bool active) {
    unsigned long long mask = __ballot(active);
    if (mask == 0) return -1;
    int lane = __lane_id();
    int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader);
    int prefix = __popcll(mask & ((1ull << lane) - 1ull));
    return active ? base + prefix : -1;
}

}  // namespace tkd
